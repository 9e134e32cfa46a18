#!/bin/bash
# TransNet on the 16x16x32 shape: tests, timing of the tile counts, per-kernel trace
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out/r05_shot_m16.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_transnet.py -x -q -m gpu > gpurun_out/r05_shot_x3_tests.txt 2>&1; tail -4 gpurun_out/r05_shot_x3_tests.txt
for cfg in SVC_SHOT_M16=3 SVC_SHOT_M16=2 SVC_SHOT_M16=0 SVC_SHOT_MX=bf16x3; do
  echo "== $cfg" >> $O
  env ${cfg//,/ } CPU=0 timeout 300 python tools/time_transnet.py 2>&1 | grep -v amdgpu.ids >> $O
done
cd /tmp && export TMPDIR=/tmp
for cfg in SVC_SHOT_M16=3; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/shot_prof
  export ${cfg//,/ }
  CPU=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/shot_prof -- python3 $GRAFT_REPO_ROOT/tools/time_transnet.py > /dev/null 2>&1
  echo "== trace $cfg" >> $O
  python3 - >> $O <<'P'
import csv, glob, os
f = max(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/shot_prof/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
seq = [(r['Kernel_Name'][:46], int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows if 'shot' in r['Kernel_Name']]
n = 12
for s in seq[-n:]: print('%-48s %8.1f us' % (s[0], s[1] / 1e3))
print('sum %.1f us' % (sum(x[1] for x in seq[-n:]) / 1e3))
P
done
cat $O
