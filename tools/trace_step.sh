#!/bin/bash
# per-dispatch timeline of ONE network pass (B=32, one stream): kernel, grid, duration, gap to the previous kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_step
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/tools/time_saliency.py > $O/run.log 2>&1
F=$(ls $O/raw/*/*kernel_trace.csv | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last complete pass: from the last k_lanczos_norm to the k_quantise after it
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(('k_lanczos', 'void k_front', 'k_front'))]
i0 = starts[-1]
i1 = next(i for i in range(i0, len(rows)) if rows[i]['Kernel_Name'].startswith('k_quantise'))
t_prev = None
tot = 0
for r in rows[i0:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if t_prev is None else (s - t_prev)
    name = r['Kernel_Name'].replace('void ', '').split('(')[0]
    wg = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1) * max(int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1), 1)
    print('%-44s wgs=%6d lds=%6s vgpr=%4s  %7.1f us  gap %5.1f' % (name[:44], wg, r.get('LDS_Block_Size', '?'), r.get('VGPR_Count', '?'), (e - s) / 1e3, gap / 1e3))
    t_prev = e
    tot += e - s
print('kernels %d  sum %.1f us  span %.1f us' % (i1 - i0 + 1, tot / 1e3, (int(rows[i1]['End_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3))
PY
rm -rf $O/raw
