#!/bin/bash
# A/B of one environment knob on the pipelined bench: tools/ab_bench.sh VAR "v1 v2 ..." [repeats]
for rep in $(seq 1 ${3:-2}); do for v in $2; do
  env $1=$v python bench.py --steps 60 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1=$v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done; done
