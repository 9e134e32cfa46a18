#!/bin/bash
# usage: tools/sweep_bench.sh "<hw queue counts>" "<pipeline depths>"
for q in $1; do for p in $2; do
  GPU_MAX_HW_QUEUES=$q python bench.py --pipeline $p --steps 60 --cpu-sample 0 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq', $q, 'P', d['config']['batches_in_flight'], d['value'], d['ms_per_step'])"
done; done
