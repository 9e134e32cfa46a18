#!/bin/bash
# round 6, job c: where a tail round goes (phase stamps of the level-bucketed Prim), bench default run with the new baseline key
mkdir -p gpurun_out
O=gpurun_out/r06_c.txt
: > $O
echo "== tail phases, bench batch (seed 100), SVC_PRIM_LVL=3 (stamps)" >> $O
SVC_PRIM_LVL=3 timeout 300 python tools/bench_map_sizes.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== without stamps" >> $O
timeout 300 python tools/bench_map_sizes.py 2>&1 | grep -v amdgpu.ids | head -4 >> $O
echo "== bench default run" >> $O
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r06_c_bench.json
python - <<'PY' >> $O
import json
d = json.loads(open('gpurun_out/r06_c_bench.json').read().strip().splitlines()[-1]); c = d['config']
print('value %.1f  ms/step %.4f' % (d['value'], d['ms_per_step']))
print(json.dumps(d['cpu_baseline'])[:900])
PY
cat $O
