"""Per-kernel table of whatever counters one rocprofv3 --pmc pass collected: python tools/pmc_generic_table.py <dir>
Counters are summed over a kernel's dispatches and printed per dispatch."""
import collections, csv, glob, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
names = []
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        c = r['Counter_Name']
        if c not in names: names.append(c)
        acc[k][c] += float(r['Counter_Value']); cnt[k][c] += 1
rows = sorted(acc.items(), key=lambda kv: -max(kv[1].values()))
print('%-42s %5s ' % ('kernel', 'disp') + ' '.join('%18s' % n[:18] for n in names))
for k, c in rows:
    n = max(cnt[k].values())
    print('%-42s %5d ' % (k[:42], n) + ' '.join('%18.1f' % (c[x] / max(cnt[k][x], 1)) for x in names))
