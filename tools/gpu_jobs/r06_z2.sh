#!/bin/bash
# round 6, job z2: k_dwpw with TWO waves per workgroup (half the footprint, twice the chain) on launches of at least N workgroups, alone / shared
mkdir -p gpurun_out
O=gpurun_out/r06_z2.txt
export SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_nwv2.so
timeout 900 python tools/time_knobs.py 4 SVC_DWPW_NWV2=1 SVC_DWPW_NWV2=400 SVC_DWPW_NWV2=1000 2>&1 | grep -v amdgpu.ids > $O
timeout 900 python tools/time_knobs.py 4 SVC_DWPW_NWV2=1 SVC_DWPW_NWV2=400 2>&1 | grep -v amdgpu.ids >> $O
cat $O
