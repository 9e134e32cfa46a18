#!/bin/bash
# round 6, job z3: the tile-shape knobs swept again on the split-bf16 pipe (round 4's sweep was on the fp32 pipe), alone / shared
mkdir -p gpurun_out
O=gpurun_out/r06_z3.txt
timeout 1200 python tools/time_knobs.py 4 SVC_PWR_NT=1 SVC_PWR_NT=3 SVC_PWR_NT=4 SVC_PWR_NT=0 SVC_PWR_NT=0,SVC_PWR_MIN_WG=256 SVC_PWR_NT=0,SVC_PWR_MIN_WG=1024 SVC_DWPW_NT=3 SVC_DWPW_NT=4 SVC_DWPW_NT=2 2>&1 | grep -v amdgpu.ids > $O
timeout 1200 python tools/time_knobs.py 4 SVC_PWR_NT=1 SVC_PWR_NT=3 SVC_PWR_NT=4 SVC_DWPW_NT=3 SVC_DWPW_NT=4 2>&1 | grep -v amdgpu.ids >> $O
cat $O
