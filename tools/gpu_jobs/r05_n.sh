#!/bin/bash
# the smoothing kernel's self-check builds beside k_pwr on the bf16 pipe (SVC_SD_EXCL=0: not alone on its CU)
mkdir -p gpurun_out
export SVC_SD_EXCL=0 SVC_MX_MASK=1
for v in ${VARIANTS:-"" 1 2}; do
  export SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_sddbg$v.so
  timeout 900 python tools/soak_network_concurrent.py 4 ${ITERS:-400} > gpurun_out/r05_sd_selfcheck$v.txt 2>&1
  echo "== variant '$v'"; grep "passes,\|self-check:" gpurun_out/r05_sd_selfcheck$v.txt
done
