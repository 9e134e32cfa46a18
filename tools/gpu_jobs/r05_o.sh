#!/bin/bash
# the packed-instruction probes beside the network with k_pwr on the bf16 pipe; the library's own kernel in its packed form
# (-DSD_PACKED build) runs in the same passes as the positive control
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/micro/pk_opsel_victim.hip -o tools/micro/libpkvictim.so 2>/dev/null
make -C retargetvid_amd/csrc OUT=../libsvc_hip_sdpacked.so EXTRA=-DSD_PACKED > /dev/null 2>&1
export SVC_SD_EXCL=0 VICTIM=${VICTIM:-2} VICTIM_LIB=$PWD/tools/micro/libpkvictim.so SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_sdpacked.so
echo "== beside k_pwr on the bf16 pipe (library built with -DSD_PACKED: its smoothing kernel is the positive control)" > gpurun_out/r05_pk_probe.txt
SVC_MX_MASK=1 timeout 900 python tools/soak_network_concurrent.py 4 ${ITERS:-600} 2>&1 | grep -v amdgpu.ids | grep -v "^iter\|^   pre\|^   got\|^   want" | tail -9 >> gpurun_out/r05_pk_probe.txt
cat gpurun_out/r05_pk_probe.txt | cut -c1-250
