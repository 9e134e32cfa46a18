#!/bin/bash
# per-launch durations of one svc_transnet_predict call (8 windows of 100 frames)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_tn
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CPU=0 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/tools/time_transnet.py > $O/run.log 2>&1
python3 - "$(ls $O/raw/*/*kernel_trace.csv | head -1)" <<'PY'
import csv, sys
rows = [dict(r, Kernel_Name=r['Kernel_Name'].replace('void ', '')) for r in csv.DictReader(open(sys.argv[1]))]
rows = [r for r in rows if r['Kernel_Name'].startswith('k_shot')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_shot_in')][-1]
tot = 0
for r in rows[last:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print('%-14s grid %8s x %-4s  %9.1f us' % (r['Kernel_Name'].split('(')[0], r['Grid_Size_X'], r['Grid_Size_Y'], d))
print('total %.1f us for 8 windows' % tot)
PY
rm -rf $O/raw
