#!/bin/bash
# stride-1 depthwise register tile (TX*10+TY); 0 = the one-output-per-thread kernel
for t in 0 21 22 41 42 44; do
  echo "== SVC_DW_TILE=$t"
  SVC_DW_TILE=$t python tools/gpu_check.py 2>&1 | grep -E "class dw|ALL OK|FAIL|rror" | head -4
  SVC_DW_TILE=$t TAG=dw$t python tools/time_saliency.py 2>&1 | tail -1
done
