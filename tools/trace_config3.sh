#!/bin/bash
# how busy is the GPU during a config-3 run?  (arguments are passed to tools/run_config3.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_config3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/tools/run_config3.py "$@" > $O/run.log 2>&1
python3 $R/tools/trace_concurrency.py $O/raw; python3 $R/tools/trace_streams.py $O/raw
grep -v "^[EWI]2026" $O/run.log | tail -1 | cut -c150-330
rm -rf $O/raw
