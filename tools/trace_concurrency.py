"""Concurrency of a pipelined bench run from a rocprofv3 kernel trace: python tools/trace_concurrency.py <dir>
Over the last 60 % of the run: wall time, share of it with 0 / 1 / 2 / 3+ kernels executing, mean number of kernels in flight."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * 4 // 10
ev = []
for s, e, _ in rows:
    if e <= lo: continue
    ev.append((max(s, lo), 1)); ev.append((e, -1))
ev.sort()
hist = {}
cur, last = 0, lo
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d; last = t
wall = t1 - lo
print('wall %.2f ms' % (wall / 1e6))
for k in sorted(hist):
    print('  %d kernels in flight: %5.1f %%' % (k, 100.0 * hist[k] / wall))
print('mean kernels in flight %.2f' % (sum(k * v for k, v in hist.items()) / wall))
tail = ('k_prim', 'k_tail_front', 'k_tail_back', 'k_prim_lvl', 'k_tree_par', 'k_tree', 'k_core', 'k_finish', 'k_sort', 'k_compact', 'k_blend', 'k_threshold')
tt = sum(e - max(s, lo) for s, e, n in rows if e > lo and n.startswith(tail))
nt = sum(e - max(s, lo) for s, e, n in rows if e > lo and not n.startswith(tail))
print('summed kernel time: tail %.1f ms, network %.1f ms (per wall ms: %.2f, %.2f)' % (tt / 1e6, nt / 1e6, tt / wall, nt / wall))
