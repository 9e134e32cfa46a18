#!/bin/bash
# round 6, job z7: batches in flight (--pipeline 4 / 5 / 6 / 8) at the driver's flags and at 100 steps, two alternations
mkdir -p gpurun_out
O=gpurun_out/r06_z7.txt
: > $O
export BENCH_CONFIG3=0 BENCH_VARIANT=0 BENCH_TORCH_BASELINE=0
for i in 1 2; do
  for STEPS in 20 100; do
    for P in 4 5 6 8; do
      python3 bench.py --steps $STEPS --warmup 5 --cpu-sample 0 --repeats 9 --pipeline $P --iso-steps 1 2>/dev/null | tail -1 | P=$P S=$STEPS python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pipeline %s steps %3s  %9.1f frames/s  %.4f ms/step  phases %s' % (os.environ['P'], os.environ['S'], d['value'], d['ms_per_step'], d['config']['batch_phase_ms_in_the_pipeline']))" >> $O
    done
  done
done
cat $O
