"""Whole-run totals of one rocprofv3 --pmc pass over bench.py: python tools/pmc_bench_totals.py <dir> <steps> <ms_per_step>
Prints MFMA-busy, VALU and LDS issue shares of the available SIMD time (1024 SIMDs x ms_per_step x clock)."""
import csv, glob, json, sys, collections
d, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
tot = collections.defaultdict(float)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r['Counter_Name']] += float(r['Counter_Value'])
out = {k: v for k, v in tot.items()}
print(json.dumps(out, indent=1))
