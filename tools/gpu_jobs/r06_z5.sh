#!/bin/bash
# round 6, job z5: k_pwr tiles by a floor of 1 024 workgroups against the default (two tiles) in the bench: driver's flags and 100 steps, three alternations
mkdir -p gpurun_out
O=gpurun_out/r06_z5.txt
: > $O
export BENCH_CONFIG3=0 BENCH_VARIANT=0 BENCH_TORCH_BASELINE=0
run() {  # $1 = label, $2.. = env
  env "${@:2}" python3 bench.py --steps $STEPS --warmup 5 --cpu-sample 0 --repeats 9 2>/dev/null | tail -1 | L="$1" S=$STEPS python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-10s steps %3s  %9.1f frames/s  %.4f ms/step  pw %.4f ms frac %.4f  one batch %.3f ms' % (os.environ['L'], os.environ['S'], d['value'], d['ms_per_step'], r['class_ms_per_step_all']['pw'], r['frac'], d['config']['one_batch_in_flight']['latency_ms_per_batch']))" >> $O
}
for i in 1 2 3; do
  for STEPS in 20 100; do
    run default X=1
    run floor1024 SVC_PWR_NT=0 SVC_PWR_MIN_WG=1024
  done
done
cat $O
