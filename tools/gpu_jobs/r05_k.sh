#!/bin/bash
# full GPU suite + bench with the TransNet x3 path and the shot planner
mkdir -p gpurun_out
S0=$SECONDS
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputest_k.txt 2>&1; tail -6 gpurun_out/r05_gputest_k.txt
echo "pytest wall $((SECONDS-S0)) s"
S0=$SECONDS
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_k.json 2> gpurun_out/r05_bench_k.err
echo "bench wall $((SECONDS-S0)) s"
python - <<'P'
import json
d = json.loads(open('gpurun_out/r05_bench_k.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('config3', 'config3_host_fed', 'config3_shot_net', 'config3_shot_net_bf16x3'):
    c = d['config'].get(k) or {}
    print(k, c.get('seconds'), c.get('seconds_all_runs'), c.get('shot_net_matrix_pipe'), c.get('error'), (c.get('scheduler_rank0') or {}).get('feeder_seconds'), c.get('windows_crc32'))
P
tail -3 gpurun_out/r05_bench_k.err
