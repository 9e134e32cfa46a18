"""Per-kernel SQ counter table from one rocprofv3 --pmc pass:
  python tools/pmc_sq_table.py <dir> [out.json]
Counters (one pass, 8 SQ slots): SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES.  WAVE_CYCLES / WAIT_* / ACTIVE_* count quad-cycles summed
over waves; their ratios are what matters: parked (s_waitcnt, barrier), issue-stalled, issuing.  MFMA busy is in cycles
summed over the SIMDs; divided by SQ_BUSY_CYCLES (per SE, summed) it is only comparable between kernels, so the table
also prints MFMA busy cycles per dispatch next to the dispatch's algorithmic MFMA cycles where known."""
import collections, csv, glob, json, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void ', '').split('(')[0]
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            cnt[name] += 1
rows = []
for name, c in acc.items():
    wc = c.get('SQ_WAVE_CYCLES', 0.0) or 1.0
    rows.append(dict(kernel=name, dispatches=cnt[name], wave_quadcycles=c.get('SQ_WAVE_CYCLES', 0),
                     parked=round(c.get('SQ_WAIT_ANY', 0) / wc, 3), issue_stalled=round(c.get('SQ_WAIT_INST_ANY', 0) / wc, 3),
                     issuing=round(c.get('SQ_ACTIVE_INST_ANY', 0) / wc, 3), valu=round(c.get('SQ_ACTIVE_INST_VALU', 0) / wc, 3),
                     lds=round(c.get('SQ_ACTIVE_INST_LDS', 0) / wc, 3),
                     mfma_busy_per_sq_busy=round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (c.get('SQ_BUSY_CYCLES', 0) or 1.0), 4),
                     mfma_busy_cycles_per_dispatch=round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(cnt[name], 1))))
rows.sort(key=lambda r: -r['wave_quadcycles'])
print('%-28s %5s %8s %8s %8s %8s %8s %10s %14s' % ('kernel', 'disp', 'parked', 'stalled', 'issuing', 'valu', 'lds', 'mfma/busy', 'mfma cyc/disp'))
for r in rows:
    print('%-28s %5d %8.3f %8.3f %8.3f %8.3f %8.3f %10.4f %14d' % (r['kernel'][:28], r['dispatches'], r['parked'], r['issue_stalled'],
                                                                  r['issuing'], r['valu'], r['lds'], r['mfma_busy_per_sq_busy'],
                                                                  r['mfma_busy_cycles_per_dispatch']))
if len(sys.argv) > 2:
    json.dump(dict(note='rocprofv3 --pmc (one pass, SQ block), network only: tools/time_saliency.py', kernels=rows), open(sys.argv[2], 'w'), indent=1)
