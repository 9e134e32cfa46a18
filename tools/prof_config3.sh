#!/bin/bash
# rocprofv3 kernel stats of a config-3 run (arguments are passed on to tools/run_config3.py): whose kernels fill the GPU?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_config3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/tools/run_config3.py "$@" > $O/run.log 2>&1
cp $(ls $O/raw/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/raw
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.1f ms'%(tot/1e6))
for r in rows[:25]:
    print('%-70s calls=%6s avg_us=%8.1f tot_ms=%8.2f pct=%s'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
tail -3 $O/run.log | cut -c1-400
