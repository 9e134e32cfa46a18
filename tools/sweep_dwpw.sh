#!/bin/bash
for w in 0 1024 4096; do
  SVC_DWPW_WIDE=$w TAG=wide$w python tools/time_saliency.py 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp && SVC_DWPW_WIDE=4096 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace7 -- python3 $GRAFT_REPO_ROOT/tools/time_saliency.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT && python tools/trace_last_pass.py gpurun_out/trace7 | tail -1 | tr '|' '\n' | grep dwpw | tr '\n' ' '
