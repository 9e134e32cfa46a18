#!/bin/bash
# round 6, job v: `bench.py --gpus 2` (self-launch, two ranks sharing the box's one GPU over gloo: BENCH_SHARE_GPU=1) -- the new test,
# then the line itself for profiles/
mkdir -p gpurun_out
O=gpurun_out/r06_v.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_dist_rccl.py -x -q 2>&1 | tail -15 >> $O
( time BENCH_SHARE_GPU=1 BENCH_VARIANT=0 BENCH_TORCH_BASELINE=0 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 --cpu-sample 0 2> gpurun_out/r06_v_bench.err | tail -1 > gpurun_out/r06_v_bench.json ) 2>> $O
tail -5 gpurun_out/r06_v_bench.err | grep -v amdgpu.ids >> $O
python - <<'PY' >> $O
import json
d = json.loads(open('gpurun_out/r06_v_bench.json').read().strip().splitlines()[-1]); c = d['config']
print('n_gpus %d value %.1f  ms/step %.4f  per rank %s  world %s' % (d['n_gpus'], d['value'], d['ms_per_step'], c['per_rank_frames_per_s'], c['world_size_seen_by_rccl']))
print(c['ranks_share_gpus'])
print('config3', c['config3']['seconds'], c['config3']['seconds_all_runs'], c['config3']['windows_crc32'], c['config3']['n_gpus'])
PY
cat $O
