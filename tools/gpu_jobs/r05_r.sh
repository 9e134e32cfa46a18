#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_transnet.py tests/test_gpu_scheduler.py -x -q -m gpu > gpurun_out/r05_shot_x3_tests.txt 2>&1; tail -6 gpurun_out/r05_shot_x3_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_k.json 2> gpurun_out/r05_bench_k.err
python - <<'P'
import json
d = json.loads(open('gpurun_out/r05_bench_k.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('config3', 'config3_host_fed', 'config3_shot_net', 'config3_shot_net_bf16x3'):
    c = d['config'].get(k) or {}
    print(k, c.get('seconds'), c.get('seconds_all_runs'), c.get('shot_net_matrix_pipe'), c.get('error'), (c.get('scheduler_rank0') or {}).get('feeder_seconds'), c.get('windows_crc32'))
P
