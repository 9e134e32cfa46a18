#!/bin/bash
# round 6, job r: phase stamps of the Prim at the benchmark's map sizes with eight workers
mkdir -p gpurun_out
SVC_PRIM_LVL=3 SIGMA=30,44 N_BLOBS=2 timeout 300 python tools/bench_map_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_r.txt
cat gpurun_out/r06_r.txt
