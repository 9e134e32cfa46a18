#!/bin/bash
for q in default 24 32; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for p in 4 8; do
    BENCH_CONFIG3=0 BENCH_VARIANT=0 python bench.py --pipeline $p --steps 60 --warmup 8 --cpu-sample 0 --repeats 7 --iso-steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q pipeline $p: %.1f fps  %.4f ms/step' % (d['value'], d['ms_per_step']))"
  done
  BENCH_VARIANT=0 BENCH_CONFIG3_EXTRA=0 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q: driver flags %.1f fps %.4f ms; config3 %s' % (d['value'], d['ms_per_step'], d['config']['config3']['seconds_all_runs']))"
done
