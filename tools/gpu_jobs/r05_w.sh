#!/bin/bash
# which co-runner triggers the packed smoothing kernel: the fp32 network + packed smoothing kernel beside TransNet's bf16 cells
mkdir -p gpurun_out
make -C retargetvid_amd/csrc OUT=../libsvc_hip_sdpacked.so EXTRA=-DSD_PACKED > /dev/null 2>&1
export SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_sdpacked.so
O=gpurun_out/r05_sd_corunner.txt
: > $O
run() { echo "== $*" >> $O; env "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^iter\|^   " | tail -3 >> $O; }
run SVC_MX=f32 SHOT=2 timeout 900 python tools/soak_network_concurrent.py 4 600
run SVC_MX=f32 timeout 900 python tools/soak_network_concurrent.py 4 600
run SVC_MX_MASK=1 timeout 900 python tools/soak_network_concurrent.py 4 600
run SVC_MX_MASK=4 timeout 900 python tools/soak_network_concurrent.py 4 600
run SVC_MX_MASK=2 timeout 900 python tools/soak_network_concurrent.py 4 600
cat $O
