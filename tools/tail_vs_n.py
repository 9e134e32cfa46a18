#!/usr/bin/env python3
"""Time of one tail round (32 maps of 140 x 250, one workgroup each) against the number of points per map, from the bench's
~2 k to the whole map: the cliff above 8 192 points that round 3's verdict named (item 4) and what replaced it.
  python tools/tail_vs_n.py  -> gpurun_out/r04_tail_vs_N.txt (copy to profiles/)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from retargetvid_amd import ops, synth, weights
from oracle import pipeline_ref as P
eng = ops.Engine(weights.make_synthetic_state_dict(0))
CP = P.init_crop_params()
rows = []
for sigma, thr in (((30, 44), 120), ((60, 90), 120), ((90, 110), 120), ((120, 150), 120), ((150, 200), 120), ((170, 240), 120), ((200, 300), 120),
                   ((200, 300), 60), ((200, 300), 1)):
    fr = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=100, n_blobs=2, sigma=sigma)).cuda()
    maps = eng.saliency(eng.resize_frames(fr, 140, 250))
    eng.threshold_(maps, thr)
    n = (maps != 0).flatten(1).sum(1).cpu().numpy()
    fl = np.zeros(32, np.uint8)
    for _ in range(2):
        m = maps.clone(); eng.cluster_center_(m, fl, CP)
    torch.cuda.synchronize(); t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        m = maps.clone(); eng.cluster_center_(m, fl, CP)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    st = eng.cluster_state(0, 35000)
    h = st['hdr']
    rows.append('N %5d .. %5d (mean %5.0f)  round of 32 maps %8.3f ms | map 0 (N = %5d): Prim %8.1f us (%4d rounds, %3d rises), sort %7.1f us, hierarchy %7.1f us (by %s), %4d clusters' % (
        n.min(), n.max(), n.mean(), ms, st['n'], h[12] / 100.0, h[16], h[17], h[8] / 100.0, h[10] / 100.0, 'k_tree_par' if h[23] else 'k_tree', h[4]))
    print(rows[-1], flush=True)
out = os.path.join(ROOT, 'gpurun_out', 'r04_tail_vs_N.txt')
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, 'w') as fp:
    fp.write('tools/tail_vs_n.py: one svc_cluster_center call over 32 maps of 140 x 250 (default parameters, no blend flags), MI355X\n' + '\n'.join(rows) + '\n')
