#!/bin/bash
# experiment: phase stamps with the probe's barrier wait separated (phase "mark" column = the wait at the mid-round barrier)
mkdir -p gpurun_out
SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_stamp.so SVC_PRIM_LVL=3 SIGMA=30,44 N_BLOBS=2 timeout 300 python tools/bench_map_sizes.py 2>&1 | grep "k_prim_lvl\|k_prim " | head -8 > gpurun_out/r06_u.txt
timeout 120 python tools/clock_under_load.py 2>&1 | grep -v amdgpu.ids | tail -5 >> gpurun_out/r06_u.txt
cat gpurun_out/r06_u.txt
