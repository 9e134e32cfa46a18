"""Per-batch timeline of a pipelined bench run from a rocprofv3 kernel trace:  python tools/trace_batches.py <dir>
Kernels are grouped by queue (= the batch's stream) and cut at k_cv_resize (first kernel of a batch).  For the batches of
the second half of the run: network span (resize .. quantise), summed network kernel durations, idle inside the span,
tail span (threshold .. last tail kernel), summed tail durations; and how many OTHER batches' network spans overlap it."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
byq = {}
for r in rows:
    byq.setdefault(r[qkey], []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').split('(')[0]))
batches = []
for q, ks in byq.items():
    ks.sort()
    cur = None
    for s, e, n in ks:
        if n.startswith('k_cv_resize'):
            cur = dict(q=q, k=[])
            batches.append(cur)
        if cur is not None:
            cur['k'].append((s, e, n))
TAIL = ('k_threshold', 'k_compact', 'k_core', 'k_prim', 'k_tail_front', 'k_tail_back', 'k_prim_lvl', 'k_tree_par', 'k_sort', 'k_tree', 'k_finish', 'k_blend')
out = []
for b in batches:
    net = [k for k in b['k'] if not k[2].startswith(TAIL)]
    tail = [k for k in b['k'] if k[2].startswith(TAIL)]
    if not net or not tail or not any(k[2].startswith('k_quantise') for k in net):
        continue
    out.append(dict(q=b['q'], n0=net[0][0], n1=net[-1][1], nsum=sum(e - s for s, e, _ in net), t0=tail[0][0], t1=tail[-1][1],
                    tsum=sum(e - s for s, e, _ in tail), nk=len(net), tk=len(tail)))
out.sort(key=lambda d: d['n0'])
half = out[len(out) // 2:]
print('%d batches (second half: %d); times in ms' % (len(out), len(half)))
print('  start    net span  net sum  net idle | tail span  tail sum | batch latency | other networks overlapping the span (fraction of it)')
for d in half[:24]:
    ov = sum(max(0, min(d['n1'], o['n1']) - max(d['n0'], o['n0'])) for o in out if o is not d)
    print('%8.2f  %8.3f %8.3f %8.3f | %8.3f %8.3f | %8.3f | %.2f' % ((d['n0'] - half[0]['n0']) / 1e6, (d['n1'] - d['n0']) / 1e6, d['nsum'] / 1e6,
          (d['n1'] - d['n0'] - d['nsum']) / 1e6, (d['t1'] - d['t0']) / 1e6, d['tsum'] / 1e6, (d['t1'] - d['n0']) / 1e6, ov / max(d['n1'] - d['n0'], 1)))
n = len(half)
print('mean: net span %.3f  net sum %.3f  tail span %.3f  tail sum %.3f  latency %.3f  step %.3f' % (
    sum(d['n1'] - d['n0'] for d in half) / n / 1e6, sum(d['nsum'] for d in half) / n / 1e6, sum(d['t1'] - d['t0'] for d in half) / n / 1e6,
    sum(d['tsum'] for d in half) / n / 1e6, sum(d['t1'] - d['n0'] for d in half) / n / 1e6, (half[-1]['n0'] - half[0]['n0']) / max(n - 1, 1) / 1e6))
