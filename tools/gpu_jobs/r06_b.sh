#!/bin/bash
# round 6, job b: reproducibility soaks at the other geometries (advisor), 12-lane jobs, PyTorch-ROCm baseline, checks of the SVC_NO_PK build
mkdir -p gpurun_out
O=gpurun_out/r06_b.txt
: > $O
run() { echo "== $*" >> $O; env "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^   " | tail -4 >> $O; }
run GEOM=187x250 timeout 600 python tools/soak_network_concurrent.py 4 500
run GEOM=250x140 timeout 600 python tools/soak_network_concurrent.py 4 500
run GEOM=187x250 timeout 600 python tools/soak_network_concurrent.py 8 250
run GEOM=250x140 timeout 600 python tools/soak_network_concurrent.py 8 250
run timeout 600 python tools/soak_network_concurrent.py 4 500
run LANES=12,12,12 timeout 900 python tools/soak_job_repeat.py 100 6
echo "== the reference's formulation under PyTorch-ROCm" >> $O
timeout 900 python tools/torch_rocm_baseline.py 32 20 2>&1 | grep -v amdgpu.ids | tail -6 >> $O
echo "== SVC_NO_PK build: kernel families, front, parity" >> $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 >> $O
BENCH_CONFIG3=0 BENCH_VARIANT=0 timeout 300 python bench.py --steps 120 --warmup 12 --cpu-sample 0 --repeats 7 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 120 steps: %.1f frames/s %.4f ms' % (d['value'], d['ms_per_step']))" >> $O
cat $O
