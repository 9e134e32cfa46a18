#!/bin/bash
# after a TransNet change: its tests, its timing, and the two bench lines kept under profiles/
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_transnet.py -x -q -m gpu > gpurun_out/r05_shot_x3_tests.txt 2>&1; tail -3 gpurun_out/r05_shot_x3_tests.txt
CPU=0 python tools/time_transnet.py 2>&1 | grep -v amdgpu.ids
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_line_driver_flags.json 2> gpurun_out/bench.err
python bench.py --steps 100 --warmup 10 --cpu-sample 32 > gpurun_out/r05_bench_line.json 2>> gpurun_out/bench.err
python - <<'P'
import json
for f in ('gpurun_out/r05_bench_line_driver_flags.json', 'gpurun_out/r05_bench_line.json'):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], [ (k, (d['config'].get(k) or {}).get('seconds')) for k in ('config3', 'config3_host_fed', 'config3_shot_net', 'config3_shot_net_bf16x3')])
P
