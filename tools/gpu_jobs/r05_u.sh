#!/bin/bash
# k_irb's split-bf16 instances at three waves per SIMD (no spills) against the default, alternating
for rep in 1 2 3; do
  echo "default:  $(python tools/time_knobs.py 4 2>/dev/null | tail -1)"
  echo "irb3:     $(SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_irb3.so python tools/time_knobs.py 4 2>/dev/null | tail -1)"
done
