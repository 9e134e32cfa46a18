#!/bin/bash
# round 6, job a: GPU suite on the ABI-5 build; packed-f32 A/B; TransNet with kept rows; driver-flag bench line
mkdir -p gpurun_out
O=gpurun_out/r06_a.txt
: > $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_a_pytest.txt
tail -3 gpurun_out/r06_a_pytest.txt >> $O
echo "== TransNet, kept rows only (r05: 94 070 frames/s)" >> $O
CPU=0 timeout 300 python tools/time_transnet.py 2>&1 | grep -v amdgpu.ids >> $O
SVC_SHOT_MX=bf16x3 CPU=0 timeout 300 python tools/time_transnet.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== packed f32 on / off, pipelined bench 120 steps (two alternations)" >> $O
BENCH_ARGS="--repeats 7" BENCH_CONFIG3=0 BENCH_VARIANT=0 timeout 900 bash tools/ab_bench_libs.sh 2 libsvc_hip.so libsvc_hip_nopk.so 2>&1 | grep -v amdgpu.ids >> $O
echo "== network pass alone / shared x4: default lib, then no-packed lib" >> $O
timeout 300 python tools/time_knobs.py 4 2>&1 | grep -v amdgpu.ids >> $O
SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_nopk.so timeout 300 python tools/time_knobs.py 4 2>&1 | grep -v amdgpu.ids >> $O
echo "== bench, driver's flags" >> $O
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_a_bench.json
python - <<'PY' >> $O
import json
d = json.loads(open('gpurun_out/r06_a_bench.json').read().strip().splitlines()[-1]); c = d['config']
print('value %.1f  ms/step %.4f' % (d['value'], d['ms_per_step']))
for k in ('config3', 'config3_host_fed', 'config3_shot_net', 'config3_shot_net_bf16x3'):
    if k in c: print(k, c[k]['seconds'], c[k]['seconds_all_runs'], c[k].get('scheduler_rank0', {}).get('plan_high_water'))
print('roofline frac', d['roofline']['frac'], 'class ms', d['roofline']['class_ms_per_step'])
PY
cat $O
