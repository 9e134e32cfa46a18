"""Soak run of the tail against the oracle: many random small maps (blobs clipped by the borders, speckle, stripes, isolated
points, grey gradients), random parameter sets and blend flags; everything must be bit-exact (maps after the filter and the
CLOSE, centres).  python tools/soak_tail.py [trials] [seed]   (GPU box; ~0.15 s per trial; SOAK_HW=140x250: every map at that size)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pipeline_ref as P, tail_ref as T
from retargetvid_amd import ops
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
eng = ops.Engine(seed=0)
bad = 0
for trial in range(trials):
    h, w = (int(rng.randint(20, 120)), int(rng.randint(20, 160))) if not os.environ.get('SOAK_HW') else tuple(int(v) for v in os.environ['SOAK_HW'].split('x'))
    mcs = int(rng.choice([3, 5, 12, 26])); ms = rng.choice([None, 2, 3, 10]); ms = None if ms is None else int(ms)
    CP = dict(P.init_crop_params(), hdbscan_min=mcs, hdbscan_min_samples=ms, select_sum=int(rng.choice([1, 2])), op_close=bool(rng.rand() < 0.8))
    n = 6
    maps = np.zeros((n, h, w), np.uint8)
    ys, xs = np.mgrid[0:h, 0:w]
    for i in range(n):
        m = rng.rand(h, w) < rng.choice([0.0, 0.005, 0.03, 0.15])
        for _ in range(rng.randint(0, 4)):
            cy, cx, ry, rx = rng.randint(-4, h + 4), rng.randint(-4, w + 4), rng.randint(2, 20), rng.randint(2, 30)
            m |= (((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2) < 1
        grad = (121 + (xs * 3 + ys * 5 + rng.randint(0, 50)) % 135).astype(np.uint8)
        vals = grad if rng.rand() < 0.5 else rng.randint(121, 256, (h, w)).astype(np.uint8)
        maps[i] = np.where(m, vals, rng.randint(0, 120, (h, w))).astype(np.uint8)
    flags = (rng.rand(n) < 0.3).astype(np.uint8)
    ref = maps.copy()                                   # the oracle's loop: threshold -> (filter, blend into the next map) -> centres
    T.threshold(ref, CP['t_threshold'])
    hwn = np.ascontiguousarray(np.transpose(ref, (1, 2, 0)))
    for i in range(n):
        hwn[:, :, i] = T.clustering_filt(hwn[:, :, i], CP, {})
        if i + 1 < n and flags[i]:
            hwn[:, :, i + 1] = T.blend_next(hwn[:, :, i], hwn[:, :, i + 1])
    dx, dy = T.centers(hwn, CP)
    ref_maps = np.transpose(hwn, (2, 0, 1))
    dm = torch.from_numpy(maps.copy()).cuda()
    eng.threshold_(dm, CP['t_threshold'])
    xy, _ = eng.cluster_center_(dm, flags, CP, want_stats=True)
    got, xy = dm.cpu().numpy(), xy.cpu().numpy()
    ok = np.array_equal(got, ref_maps)
    for i in range(n):
        if dx[i] is None: ok = ok and bool(np.isnan(xy[i]).all())
        else: ok = ok and xy[i, 0] == dx[i] and xy[i, 1] == dy[i]
    if not ok:
        bad += 1
        print('MISMATCH trial %d: %dx%d mcs=%d ms=%s select_sum=%d close=%s' % (trial, h, w, mcs, ms, CP['select_sum'], CP['op_close']), flush=True)
print('%d trials, %d mismatches' % (trials, bad))
sys.exit(1 if bad else 0)
