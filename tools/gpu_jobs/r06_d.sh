#!/bin/bash
# round 6, job d: the pruned build (TransNet losers, k_stem FMA form, SVC_SD_EXCL, SVC_TAIL_PRIO gone): GPU suite, smoke, TransNet rate, bench
mkdir -p gpurun_out
O=gpurun_out/r06_d.txt
: > $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_d_pytest.txt
tail -3 gpurun_out/r06_d_pytest.txt >> $O
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
CPU=0 timeout 300 python tools/time_transnet.py 2>&1 | grep -v amdgpu.ids >> $O
BENCH_CONFIG3_EXTRA=0 BENCH_VARIANT=0 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench driver flags: %.1f frames/s %.4f ms; config3 %s' % (d['value'], d['ms_per_step'], d['config']['config3']['seconds_all_runs']))" >> $O
cat $O
