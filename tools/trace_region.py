"""Timeline of the LAST timed region of a bench run from a rocprofv3 kernel trace: python tools/trace_region.py <dir> <steps>
Every batch of the region: queue, network start / end, tail end (ms from the region's first kernel); then the region's span
against steps x the steady step -- what the pipeline's fill and drain cost at the driver's 20 steps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
steps = int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
byq = {}
for r in rows:
    byq.setdefault(r['Queue_Id'], []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').split('(')[0]))
batches = []
for q, ks in byq.items():
    ks.sort()
    cur = None
    for s, e, n in ks:
        if n.startswith('k_cv_resize'):
            cur = dict(q=q, k=[]); batches.append(cur)
        if cur is not None: cur['k'].append((s, e, n))
batches = [b for b in batches if any(k[2].startswith('k_quantise') for k in b['k'])]
batches.sort(key=lambda b: b['k'][0][0])
reg = batches[-steps:]
t0 = reg[0]['k'][0][0]
end = max(k[1] for b in reg for k in b['k'])
# kernels of the region's queues after the last batch start (flush rounds) belong to the region too
print('queue  net start  net end  tail end   (ms)')
for b in reg:
    net_end = max(k[1] for k in b['k'] if k[2].startswith('k_quantise'))
    print('%5s  %8.3f %8.3f %8.3f' % (b['q'], (b['k'][0][0] - t0) / 1e6, (net_end - t0) / 1e6, (b['k'][-1][1] - t0) / 1e6))
print('region span %.3f ms for %d steps = %.4f ms per step' % ((end - t0) / 1e6, steps, (end - t0) / 1e6 / steps))
mid = reg[steps // 4: steps - steps // 4]
if len(mid) > 2:
    print('steady step (batch starts of the middle half): %.4f ms' % ((mid[-1]['k'][0][0] - mid[0]['k'][0][0]) / 1e6 / (len(mid) - 1)))
