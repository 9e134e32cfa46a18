#!/bin/bash
# TransNet x3: per-kernel times for SVC_SHOT_PT=1 / 2, and |dP| of the pipes against the oracle
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out/r05_shot_x3_b.txt
: > $O
python - >> $O 2>&1 <<'P'
import os, numpy as np, torch, sys
sys.path.insert(0, '.')
from retargetvid_amd import transnetv1_handler as Hd, weights
from oracle import transnet_ref as R
sd = weights.make_transnet_state_dict(0)
rs = np.random.RandomState(3)
# smooth-ish frames with two cuts (the tests' generator is smoother; this is harsher: per-pixel noise)
fr = rs.randint(0, 256, (200, 27, 48, 3)).astype(np.uint8)
fr[:70] = (fr[:70] // 4 + 30); fr[70:140] = (fr[70:140] // 4 + 150)
ref = R.predict_video(sd, fr)
for mx in ('f32', 'bf16x6', 'bf16x3'):
    os.environ['SVC_SHOT_MX'] = mx
    net = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=sd)
    p = net.predict_video(torch.from_numpy(fr).cuda())
    p = p.cpu().numpy() if torch.is_tensor(p) else np.asarray(p)
    print('%-7s max|dP| vs oracle %.3e   mean|dP| %.3e   (P range %.3f..%.3f)' % (mx, np.abs(p - ref).max(), np.abs(p - ref).mean(), ref.min(), ref.max()))
    net.close()
P
cd /tmp && export TMPDIR=/tmp
for pt in 2 1; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/shot_prof
  SVC_SHOT_PT=$pt SVC_SHOT_MX=bf16x6 CPU=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/shot_prof -- python3 $GRAFT_REPO_ROOT/tools/time_transnet.py > /dev/null 2>&1
  echo "== PT=$pt" >> $O
  python3 - >> $O <<'P'
import csv, glob, os
f = max(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/shot_prof/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
seq = [(r['Kernel_Name'][:46], int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows if 'shot' in r['Kernel_Name']]
for s in seq[-12:]: print('%-48s %8.1f us' % s[:1] + '' if False else '%-48s %8.1f us' % (s[0], s[1] / 1e3))
print('sum %.1f us' % (sum(x[1] for x in seq[-12:]) / 1e3))
P
done
cat $O
