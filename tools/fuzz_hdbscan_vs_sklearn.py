"""Live fuzz of oracle/hdbscan_ref.py against scikit-learn's HDBSCAN (round-4 verdict, "Next round" 3b).

The committed pin (tests/golden/hdbscan_sklearn.npz, hdbscan_tieorder.npz) is 12 + 126 cases; this script widens it at test
time: N random thresholded saliency-like maps (Gaussian-blob grey maps at 140x250 thresholded at 120 with the default
parameters min_cluster_size 26 / min_samples None, and the same maps shrunk to 35x62 by cv_ref's INTER_LINEAR and
thresholded at 90 with the ISM'21 set 5 / 3 -- smartVidCrop.py:1089-1100, :2340-2348 -- plus tie-heavy adversaries: lattices,
lines, sparse noise), each clustered by (a) the oracle in numpy's edge order and (b) sklearn.cluster.HDBSCAN driven with the
min_samples + 1 mapping; labels must agree bit for bit.

Like tools/make_golden_hdbscan.py it re-runs itself with numpy's SIMD argsort dispatch disabled, so that sklearn's
np.argsort of the MST weights takes the scalar introsort of the reference's numpy 1.19.

    python tools/fuzz_hdbscan_vs_sklearn.py [n_maps] [seed]      -> one JSON line on stdout
"""
import json
import os
import subprocess
import sys
import warnings

DISABLE = 'AVX512F AVX512CD AVX512VL AVX512BW AVX512DQ AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3'
if os.environ.get('NPY_DISABLE_CPU_FEATURES') != DISABLE:
    sys.exit(subprocess.call([sys.executable] + sys.argv, env=dict(os.environ, NPY_DISABLE_CPU_FEATURES=DISABLE)))

import numpy as np                                   # noqa: E402
from numpy._core._multiarray_umath import __cpu_features__ as _feat      # noqa: E402
assert not _feat['AVX2'] and not _feat['AVX512_SKX'], 'numpy still dispatches a SIMD argsort'
from sklearn.cluster import HDBSCAN                  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cv_ref, hdbscan_ref as H          # noqa: E402

warnings.filterwarnings('ignore')


def sk_labels(X, mcs, ms):
    k = H.effective_min_samples(len(X), mcs, ms)
    return HDBSCAN(min_cluster_size=mcs, min_samples=k + 1, metric='sqeuclidean', allow_single_cluster=True,
                   algorithm='brute').fit_predict(np.asarray(X, float)).astype(np.int64)


def grey_map(r, hw=(140, 250)):
    """A saliency-like u8 map: 1-4 Gaussian blobs of random size and peak over low noise."""
    ys, xs = np.mgrid[0:hw[0], 0:hw[1]].astype(np.float64)
    m = r.uniform(0, 25, hw)
    for _ in range(r.randint(1, 5)):
        cy, cx = r.uniform(0, hw[0]), r.uniform(0, hw[1])
        sy, sx = r.uniform(2, 14), r.uniform(2, 20)
        m += r.uniform(100, 255) * np.exp(-0.5 * (((ys - cy) / sy) ** 2 + ((xs - cx) / sx) ** 2))
    return np.clip(m, 0, 255).astype(np.uint8)


def cases(n_maps, seed):
    r = np.random.RandomState(seed)
    i = 0
    while i < n_maps:
        kind = i % 10
        if kind < 4:                                   # default parameters on the full map
            m = grey_map(r)
            pts = np.argwhere(m >= 120)
            yield 'default', pts, 26, None
        elif kind < 8:                                 # ISM'21 parameters on the 35x62 map (resize_factor 4)
            m = cv_ref.resize_linear_factor_u8(grey_map(r), 0.25)
            assert m.shape == (35, 62)
            pts = np.argwhere(m >= 90)
            yield 'best', pts, 5, 3
        elif kind == 8:                                # tie-heavy: a lattice patch with holes, or crossing lines
            hw = (int(r.randint(6, 30)), int(r.randint(6, 40)))
            if r.rand() < 0.5:
                m = r.rand(*hw) < r.uniform(0.5, 1.0)
                m[::int(r.randint(2, 4))] &= r.rand(hw[1]) < 0.8
            else:
                m = np.zeros(hw, bool)
                m[hw[0] // 2] = True
                m[:, hw[1] // 3] = True
                m |= np.eye(*hw, dtype=bool)
            yield 'lattice', np.argwhere(m), int(r.choice([2, 5, 26])), (None if r.rand() < 0.5 else int(r.randint(1, 6)))
        else:                                          # sparse noise + one blob
            m = r.rand(70, 125) < r.uniform(0.005, 0.05)
            cy, cx = r.randint(10, 60), r.randint(10, 115)
            m[cy - 4:cy + 4, cx - 6:cx + 6] = True
            yield 'noise', np.argwhere(m), int(r.choice([5, 10, 26])), (None if r.rand() < 0.5 else 3)
        i += 1


def main():
    n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    done, skipped, bad, by_kind, npts = 0, 0, [], {}, []
    for kind, pts, mcs, ms in cases(n_maps, seed):
        # the reference clusters only maps with more than hdbscan_min + 1 points (smartVidCrop.py:1086-1088)
        if len(pts) <= mcs + 1 or len(pts) > 4000:
            skipped += 1
            continue
        X = pts.astype(np.int64)
        lab = H.hdbscan_labels(X, mcs, ms)
        sk = sk_labels(X, mcs, ms)
        if not np.array_equal(lab, sk):
            bad.append(dict(kind=kind, n=int(len(X)), mcs=mcs, ms=ms, case=done + skipped))
        done += 1
        by_kind[kind] = by_kind.get(kind, 0) + 1
        npts.append(len(X))
    print(json.dumps(dict(compared=done, skipped=skipped, mismatches=bad, by_kind=by_kind,
                          points_min=int(min(npts)), points_max=int(max(npts)), points_mean=float(np.mean(npts)), seed=seed)))


if __name__ == '__main__':
    main()
