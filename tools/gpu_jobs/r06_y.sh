#!/bin/bash
# round 6, job y: the end-to-end soak on the trained-like and the no-carrier checkpoints
mkdir -p gpurun_out
O=gpurun_out/r06_y.txt
: > $O
for c in "SOAK_CHECKPOINT=tl timeout 1800 python tools/soak_e2e.py 60 23" "SOAK_CHECKPOINT=nc timeout 1800 python tools/soak_e2e.py 50 24" "SOAK_CHECKPOINT=tl2 timeout 1800 python tools/soak_e2e.py 40 25"; do
  echo "== $c" >> $O
  bash -c "$c" 2>&1 | grep -v amdgpu.ids | grep -v ": identical" | tail -30 >> $O
done
cat $O
