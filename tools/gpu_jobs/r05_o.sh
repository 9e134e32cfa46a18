#!/bin/bash
# the packed-instruction probes beside the network with k_pwr on the bf16 pipe (and, as a control, beside the fp32 network)
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/micro/pk_opsel_victim.hip -o tools/micro/libpkvictim.so 2>/dev/null
export SVC_SD_EXCL=0 VICTIM=${VICTIM:-2} VICTIM_LIB=$PWD/tools/micro/libpkvictim.so
echo "== beside k_pwr on the bf16 pipe" > gpurun_out/r05_pk_probe.txt
SVC_MX_MASK=1 timeout 900 python tools/soak_network_concurrent.py 4 ${ITERS:-600} 2>&1 | grep -v amdgpu.ids | tail -8 >> gpurun_out/r05_pk_probe.txt
echo "== beside the fp32 network" >> gpurun_out/r05_pk_probe.txt
SVC_MX=f32 timeout 900 python tools/soak_network_concurrent.py 4 ${ITERS:-600} 2>&1 | grep -v amdgpu.ids | tail -8 >> gpurun_out/r05_pk_probe.txt
echo "== alone (one engine, no co-runner streams besides its own pass)" >> gpurun_out/r05_pk_probe.txt
SVC_MX=f32 timeout 900 python tools/soak_network_concurrent.py 1 100 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r05_pk_probe.txt
cat gpurun_out/r05_pk_probe.txt | cut -c1-250
