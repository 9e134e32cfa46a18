"""Per-queue view of a rocprofv3 kernel trace: python tools/trace_streams.py <dir>
For every hardware queue: kernels, busy time, share of the run, and the idle gaps between consecutive kernels (how much of the
idle time sits in gaps of < 20 us / 20-200 us / 0.2-2 ms / > 2 ms) -- short gaps are launch overhead, long ones a host that
was doing something else (or waiting)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
q = collections.defaultdict(list)
for r in rows:
    q[r.get('Queue_Id', '?')].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
lo = t0 + (t1 - t0) * 3 // 10
print('columns:', list(rows[0].keys()))
print('run %.1f ms; analysed: the last 70 %%' % ((t1 - t0) / 1e6))
for k, v in sorted(q.items(), key=lambda kv: -len(kv[1])):
    v = sorted(x for x in v if x[0] >= lo)
    if len(v) < 10: continue
    busy = sum(e - s for s, e, _ in v)
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    gaps = [g for g in gaps if g > 0]
    b = [0, 0, 0, 0]
    for g in gaps:
        b[0 if g < 20e3 else 1 if g < 200e3 else 2 if g < 2e6 else 3] += g
    span = v[-1][1] - v[0][0]
    print('queue %-4s kernels %6d  busy %5.1f %% of its span %.1f ms | idle in gaps <20us %.1f ms, 20-200us %.1f, 0.2-2ms %.1f, >2ms %.1f' % (
        k, len(v), 100.0 * busy / span, span / 1e6, b[0] / 1e6, b[1] / 1e6, b[2] / 1e6, b[3] / 1e6))
    # what runs right after the long gaps
    after = collections.Counter(v[i + 1][2].split('(')[0][:40] for i in range(len(v) - 1) if v[i + 1][0] - v[i][1] > 200e3)
    print('      kernels after gaps > 0.2 ms:', after.most_common(6))
