#!/bin/bash
# round 6, job zf: final validation of HEAD: GPU suite, smoke (build() then smoke() in one process), the driver's bench command
mkdir -p gpurun_out
O=gpurun_out/r06_zf.txt
: > $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_zf_pytest.txt
tail -3 gpurun_out/r06_zf_pytest.txt >> $O
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_zf_bench.json ) 2>> $O
python - <<'PY' >> $O
import json
d = json.loads(open('gpurun_out/r06_zf_bench.json').read().strip().splitlines()[-1]); c = d['config']
print('value %.1f  ms/step %.4f  regions %s' % (d['value'], d['ms_per_step'], c['repeats']['ms_per_step']))
for k in ('config3', 'config3_host_fed', 'config3_shot_net', 'config3_shot_net_bf16x3'):
    print(k, c[k]['seconds'], c[k]['seconds_all_runs'])
print('roofline', {k: d['roofline'][k] for k in ('frac', 'class_ms_per_step', 'frac_event_corrected', 'frac_wall', 'traffic_per_step')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['network_under_pytorch_rocm'].get('value'))
PY
cat $O
