#!/bin/bash
# round 6, job z4: k_pwr's column tiles per workgroup chosen by a workgroup floor (SVC_PWR_NT=0) at several floors, three alternations
mkdir -p gpurun_out
O=gpurun_out/r06_z4.txt
: > $O
for i in 1 2 3; do
timeout 1200 python tools/time_knobs.py 4 SVC_PWR_NT=0,SVC_PWR_MIN_WG=1024 SVC_PWR_NT=0,SVC_PWR_MIN_WG=768 SVC_PWR_NT=0,SVC_PWR_MIN_WG=1536 SVC_PWR_NT=0,SVC_PWR_MIN_WG=2048 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
