#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_region
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH_PLAIN=1 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/bench.py --steps ${STEPS:-20} --warmup 3 --repeats 3 --cpu-sample 0 "$@" > $O/run.log 2>&1
python3 $R/tools/trace_region.py $O/raw ${STEPS:-20}
tail -1 $O/run.log | cut -c1-200
rm -rf $O/raw
