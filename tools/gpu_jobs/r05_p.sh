#!/bin/bash
# the scalar bilinear without the CU isolation (SVC_SD_EXCL=0): network soaks, pipeline soak, job soak
mkdir -p gpurun_out
O=gpurun_out/r05_sd_fix_soak.txt
: > $O
export SVC_SD_EXCL=0
run() { echo "== $*" >> $O; env "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-3} >> $O; }
run SVC_MX_MASK=1 timeout 1200 python tools/soak_network_concurrent.py 4 1500
run timeout 1200 python tools/soak_network_concurrent.py 4 1500
run timeout 1200 python tools/soak_network_concurrent.py 8 400
run timeout 1200 python tools/soak_network_concurrent.py 2 1000
run SVC_SMOOTH_MFMA=0 SVC_MX_MASK=1 timeout 1200 python tools/soak_network_concurrent.py 4 800
run timeout 1500 python tools/soak_pipeline_concurrent.py 4 400
run LANES=12 timeout 1500 python tools/soak_job_repeat.py 80 100
cat $O
