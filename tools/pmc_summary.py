"""Summarise rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <dir> [counter]"""
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')[:40]
        key = (name, r.get('Counter_Name'))
        rows[key][0] += 1
        rows[key][1] += float(r.get('Counter_Value', 0))
for (name, c), (n, v) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:30]:
    print('%-42s %-12s dispatch-rows %6d  sum %14.1f  per-dispatch %12.2f' % (name, c, n, v, v / max(n, 1)))
