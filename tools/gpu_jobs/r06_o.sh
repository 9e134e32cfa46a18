#!/bin/bash
# round 6, job o: network passes on four streams BESIDE TransNet's bf16 cells (a second kind of bf16-MFMA co-runner), three geometries; long soak at the benchmark geometry
mkdir -p gpurun_out
O=gpurun_out/r06_o.txt
: > $O
run() { echo "== $*" >> $O; env "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^   " | tail -3 >> $O; }
run SHOT=2 timeout 900 python tools/soak_network_concurrent.py 4 500
run SHOT=2 GEOM=187x250 timeout 900 python tools/soak_network_concurrent.py 4 400
run SHOT=2 GEOM=250x140 timeout 900 python tools/soak_network_concurrent.py 4 400
run timeout 1500 python tools/soak_network_concurrent.py 4 3000
run timeout 1500 python tools/soak_pipeline_concurrent.py 8 400
cat $O
