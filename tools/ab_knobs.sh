#!/bin/bash
# A/B of tuning knobs on the benchmark: per knob set, the pipelined step and the un-pipelined 'pw' class time.
#   tools/ab_knobs.sh "SVC_DWPW_NT=5" "SVC_DWPW_NT=3" ...     (a set may hold several assignments separated by spaces)
for set in "$@"; do
  env $set BENCH_CONFIG3=0 python bench.py --steps 60 --cpu-sample 0 --repeats 5 --iso-steps 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-40s %8.0f fps  %.4f ms/step  pw %.4f ms (frac %.4f)  stem %.4f  all-net %.4f  one-batch %.3f ms' % ('$set', d['value'], d['ms_per_step'], r['class_ms_per_step_all']['pw'], r['frac'], r['class_ms_per_step_all']['stem'], sum(r['class_ms_per_step_all'][k] for k in ('resize','lanczos','stem','pw','dw','resample','smooth')), d['config']['one_batch_in_flight']['latency_ms_per_batch']))"
done
