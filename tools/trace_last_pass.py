"""Print per-launch durations of the last network pass in a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_lanczos')]
seg = rows[idx[-1]:]
out = []
for r in seg:
    n = r['Kernel_Name']
    out.append((n.split('(')[0].replace('void ', '')[:22], int(r.get('Grid_Size', 0)), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
tot = sum(o[2] for o in out)
print('launches %d  sum %.1f us' % (len(out), tot))
print(' | '.join('%s:%d:%.0f' % o for o in out))
