#!/bin/bash
# TransNet cells on the split-bf16 pipe: parity tests, then timing of the forms
mkdir -p gpurun_out
O=gpurun_out/r05_shot_x3.txt
CFGS=${CFGS:-"SVC_SHOT_MX=bf16x6 SVC_SHOT_MX=bf16x6,SVC_SHOT_PT=1 SVC_SHOT_MX=bf16x6,SVC_SHOT_XCD=0 SVC_SHOT_MX=bf16x3 SVC_SHOT_MX=bf16x3,SVC_SHOT_PT=1"}
: > $O
timeout 900 python -m pytest tests/test_gpu_transnet.py -x -q -m gpu > gpurun_out/r05_shot_x3_tests.txt 2>&1; tail -5 gpurun_out/r05_shot_x3_tests.txt
for cfg in $CFGS; do
  echo "== $cfg" >> $O
  env ${cfg//,/ } CPU=0 timeout 300 python tools/time_transnet.py >> $O 2>&1
done
cat $O
cd /tmp && export TMPDIR=/tmp
SVC_SHOT_MX=bf16x6 CPU=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/shot_prof -- python3 $GRAFT_REPO_ROOT/tools/time_transnet.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls gpurun_out/shot_prof/*/*kernel_stats.csv | head -1) gpurun_out/r05_transnet_kernel_stats.csv
head -12 gpurun_out/r05_transnet_kernel_stats.csv | cut -c1-160
