#!/bin/bash
# XCD-aware workgroup order in the network kernels (-DSVC_XCD_REMAP build) against the default, alternating
for rep in 1 2 3; do
  echo "default:  $(python tools/time_knobs.py 4 2>/dev/null | tail -1)"
  echo "xcd:      $(SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_xcd.so python tools/time_knobs.py 4 2>/dev/null | tail -1)"
done
SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_xcd.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_layer or families" 2>&1 | tail -2
