"""Per-kernel mean of every counter in a rocprofv3 --pmc CSV dir, joined with the kernel trace durations."""
import csv, glob, sys, collections
d = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        vals[r['Kernel_Name'][:28] + '|' + r.get('Grid_Size', '')][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for v in vals.values() for c in v})
print('%-40s %5s ' % ('kernel|grid', 'n') + ' '.join('%14s' % c[:14] for c in names))
for k, v in sorted(vals.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', kv[1].get(names[0], [0])))):
    n = len(next(iter(v.values())))
    print('%-40s %5d ' % (k, n) + ' '.join('%14.0f' % (sum(v.get(c, [0])) / max(len(v.get(c, [0])), 1)) for c in names))
