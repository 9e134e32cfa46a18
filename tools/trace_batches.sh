#!/bin/bash
# per-batch timeline of the pipelined bench (arguments are passed to bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_batches
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH_PLAIN=1 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/bench.py --steps 40 --warmup 5 --cpu-sample 0 "$@" > $O/run.log 2>&1
python3 $R/tools/trace_batches.py $O/raw
rm -rf $O/raw
