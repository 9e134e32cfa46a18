#!/bin/bash
# A/B of library builds in the pipelined bench: ab_bench_libs.sh <repeats> <lib> [<lib> ...]   (paths relative to retargetvid_amd/)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    SVC_LIB=$R/retargetvid_amd/$L python3 bench.py --steps 120 --warmup 12 --cpu-sample 0 ${BENCH_ARGS:-} 2>/dev/null | tail -1 | L=$L python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('%-28s %9.1f frames/s  %.4f ms/step  %s' % (os.environ['L'], d['value'], d['ms_per_step'], c['batch_phase_ms_in_the_pipeline']))"
  done
done
