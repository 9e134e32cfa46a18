#!/bin/bash
# round 6, job x: the end-to-end soak with every centre difference explained (raw maps compared, oracle tail on the GPU's maps)
mkdir -p gpurun_out
O=gpurun_out/r06_x.txt
: > $O
for c in "timeout 1800 python tools/soak_e2e.py 70 21" "timeout 1800 python tools/soak_e2e.py 70 22"; do
  echo "== $c" >> $O
  bash -c "$c" 2>&1 | grep -v amdgpu.ids | tail -40 >> $O
done
cat $O
