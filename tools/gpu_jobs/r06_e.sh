#!/bin/bash
# round 6, job e: stage costs of the network pass (alone / four passes sharing the chip) on the final kernels
mkdir -p gpurun_out
timeout 900 python tools/time_segments.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_e.txt
cat gpurun_out/r06_e.txt
