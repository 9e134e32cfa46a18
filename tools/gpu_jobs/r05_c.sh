#!/bin/bash
for p in 3 4 5 6 8; do
  BENCH_CONFIG3=0 BENCH_VARIANT=0 python bench.py --pipeline $p --steps 60 --warmup 8 --cpu-sample 0 --repeats 7 --iso-steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $p: %.1f fps  %.4f ms/step  phases %s  host %s' % (d['value'], d['ms_per_step'], d['config']['batch_phase_ms_in_the_pipeline'], d['config']['host_ms_per_step']))"
done
