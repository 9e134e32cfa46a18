"""tests/golden/hdbscan_sklearn.npz: labels of scikit-learn's HDBSCAN (a port of hdbscan 0.8.x)
on seeded pixel sets, with the min_samples+1 mapping.  `exact_i` records
whether the oracle's stable edge order reproduced sklearn's labels when the golden was made
(sklearn sorts MST edges with numpy's unstable default argsort, so ties may resolve differently).
Run from the repo root:  python tools/make_golden_hdbscan.py"""
import os
import sys
import warnings

import numpy as np
from sklearn.cluster import HDBSCAN

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hdbscan_ref as H        # noqa: E402

warnings.filterwarnings('ignore')


def blob_points(seed, nb, hw=(140, 250), noise=0.002):
    r = np.random.RandomState(seed)
    m = np.zeros(hw, bool)
    ys, xs = np.mgrid[0:hw[0], 0:hw[1]]
    for _ in range(nb):
        cy, cx = r.uniform(10, hw[0] - 10), r.uniform(10, hw[1] - 10)
        sy, sx = r.uniform(3, 16), r.uniform(3, 22)
        m |= (((ys - cy) / sy) ** 2 + ((xs - cx) / sx) ** 2) < 1
    m |= r.rand(*hw) < noise
    m &= r.rand(*hw) < 0.97
    return np.argwhere(m)


out = {}
cases = [(0, 1, 26, 0), (1, 2, 26, 0), (2, 3, 26, 0), (3, 4, 26, 0), (4, 2, 5, 3), (5, 3, 5, 3), (6, 5, 26, 0),
         (7, 1, 5, 3), (8, 3, 10, 0), (9, 6, 15, 4), (10, 2, 26, 0), (11, 4, 5, 3)]
for i, (seed, nb, mcs, ms) in enumerate(cases):
    X = blob_points(seed, nb, noise=[0.002, 0.01, 0.0][seed % 3])
    k = H.effective_min_samples(len(X), mcs, ms or None)
    sk = HDBSCAN(min_cluster_size=mcs, min_samples=k + 1, metric='sqeuclidean', allow_single_cluster=True,
                 algorithm='brute').fit_predict(X.astype(float))
    mine = H.hdbscan_labels(X, mcs, ms or None)
    out['X_%d' % i] = X.astype(np.int16)
    out['params_%d' % i] = np.array([mcs, ms])
    out['labels_%d' % i] = sk.astype(np.int32)
    out['exact_%d' % i] = np.array(bool((mine == sk).all()))
    print(i, len(X), 'exact' if (mine == sk).all() else 'agree %.3f' % (mine == sk).mean())
out['n_cases'] = np.array(len(cases))
np.savez_compressed('tests/golden/hdbscan_sklearn.npz', **out)
