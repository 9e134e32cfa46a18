#!/bin/bash
# round 6, job w: soaks with new seeds -- the whole path against the oracle pipeline on random videos / parameters (tools/soak_e2e.py),
# the tail against the oracle on random maps (new seeds)
mkdir -p gpurun_out
O=gpurun_out/r06_w.txt
: > $O
for c in "timeout 1500 python tools/soak_e2e.py 70 21" "timeout 900 python tools/soak_tail.py 500 31" "SOAK_HW=140x250 timeout 600 python tools/soak_tail.py 40 32"; do
  echo "== $c" >> $O
  bash -c "$c" 2>&1 | grep -v amdgpu.ids | tail -25 >> $O
done
cat $O
