#!/bin/bash
# k_prim variants: smallest points-per-thread (fewer, fatter wavefronts per map)
for pt in 2 4 8 16; do
  echo "== SVC_PRIM_PT=$pt"
  SVC_PRIM_PT=$pt python tools/gpu_check.py 2>&1 | grep -E "class prim|warm frame  0|warm frame 28|ALL OK|FAIL|Error|error" | head -8
done
