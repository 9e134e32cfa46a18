#!/bin/bash
# round 6, job z6: one column tile per k_pwr workgroup per K (mask: 1 = K 64, 2 = 96, 4 = 128, 8 = 160; low-resolution levels), alone / shared, two alternations
mkdir -p gpurun_out
O=gpurun_out/r06_z6.txt
: > $O
export SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_nt1.so
for i in 1 2; do
timeout 1200 python tools/time_knobs.py 4 SVC_PWR_NT1_MASK=1 SVC_PWR_NT1_MASK=2 SVC_PWR_NT1_MASK=4 SVC_PWR_NT1_MASK=8 SVC_PWR_NT1_MASK=9 SVC_PWR_NT1_MASK=12 SVC_PWR_NT1_MASK=15 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
