#!/bin/bash
# round 6, job z1: k_dwpw with the K split eight ways on launches of few workgroups (SVC_DWPW_NWV8 = largest grid that takes it), alone / shared
mkdir -p gpurun_out
O=gpurun_out/r06_z1.txt
export SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_nwv8.so
timeout 900 python tools/time_knobs.py 4 SVC_DWPW_NWV8=300 SVC_DWPW_NWV8=600 SVC_DWPW_NWV8=2000 SVC_DWPW_NWV8=0 2>&1 | grep -v amdgpu.ids > $O
timeout 900 python tools/time_knobs.py 4 SVC_DWPW_NWV8=300 SVC_DWPW_NWV8=600 2>&1 | grep -v amdgpu.ids >> $O
cat $O
