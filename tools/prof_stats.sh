#!/bin/bash
# rocprofv3 kernel stats of bench.py (arguments are passed on), summary printed and left in gpurun_out/prof_stats/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_stats
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --cpu-sample 0 "$@" > $O/bench.log 2>&1
cp $(ls $O/raw/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/raw
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:40]:
    print('%-70s calls=%5s avg_us=%8.1f tot_ms=%8.2f pct=%s'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
tail -1 $O/bench.log | cut -c1-300
