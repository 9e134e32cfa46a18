#!/bin/bash
# pipelined step against the number of batches in flight (and the calls a stream may have outstanding)
for p in "$@"; do
  BENCH_CONFIG3=0 python bench.py --steps ${STEPS:-100} --cpu-sample 0 --repeats ${REPEATS:-5} --iso-steps 1 --pipeline $p 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pipeline %s: %8.0f fps  %.4f ms/step  phases %s  host %s' % ('$p', d['value'], d['ms_per_step'], d['config']['batch_phase_ms_in_the_pipeline'], d['config']['host_ms_per_step']))"
done
