"""tests/golden/loess_golden.npz: outputs of the reference's OWN LOESS (3rd_party_libs/loess/pyloess.py:13-95,
imported here in the build container: it needs numpy only) driven exactly as loess_handler drives it
(smartVidCrop.py:1629-1646: x = 0..n-1, one Loess.estimate(j, window, use_matrix=False, degree) per frame),
plus the whole sc_smoothing chain restated around it for one multi-shot track (Butterworth + LOESS with the
reference's window rule, smartVidCrop.py:1648-1734).  The vectors pin oracle/temporal_ref.py and
retargetvid_amd/temporal.py; /root/reference is not needed to run the tests.
Run from the repo root:  python tools/make_golden_loess.py"""
import importlib.util
import os

import numpy as np

spec = importlib.util.spec_from_file_location('pyloess', '/root/reference/3rd_party_libs/loess/pyloess.py')
pyloess = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pyloess)


def ref_handler(di, window, degree):
    """loess_handler with loess_filt=1 (smartVidCrop.py:1629-1641), including the NaN fall-back."""
    cl = len(di)
    if cl < 10:
        return np.array(di, float)
    lo = pyloess.Loess(np.arange(cl), np.asarray(di, float))
    ds = [lo.estimate(j, window=window, use_matrix=False, degree=degree) for j in range(cl)]
    if np.isnan(np.sum(ds)):
        ds = list(di)
    return np.array(ds, float)


rng = np.random.RandomState(7)
out = {}
cases = []
# (length, window, degree, kind)
for n, w, d, kind in [(10, 7, 2, 'walk'), (12, 9, 1, 'walk'), (37, 35, 2, 'walk'), (90, 59, 2, 'walk'), (90, 59, 1, 'walk'),
                      (150, 59, 2, 'steps'), (61, 59, 2, 'sine'), (300, 49, 2, 'walk'), (300, 49, 3, 'sine'),
                      (45, 43, 2, 'const'), (9, 7, 2, 'walk'), (200, 59, 2, 'ints'), (64, 5, 2, 'walk'), (33, 31, 1, 'steps'),
                      (50, 8, 2, 'walk'), (120, 58, 2, 'sine'), (75, 30, 1, 'walk')]:      # even windows: never produced by sc_smoothing (:1668-1670), pinned anyway
    t = np.arange(n)
    if kind == 'walk':
        y = 125 + np.cumsum(rng.randn(n) * 2.0)
    elif kind == 'steps':
        y = np.repeat(rng.uniform(20, 230, (n + 19) // 20), 20)[:n] + rng.randn(n) * 0.3
    elif kind == 'sine':
        y = 70 + 40 * np.sin(t / 9.0) + rng.randn(n)
    elif kind == 'ints':
        y = np.round(125 + 60 * np.sin(t / 31.0) + rng.randn(n) * 3).astype(float)
    else:
        y = np.full(n, 88.5)           # constant track: normalize_array divides by zero -> NaN -> fall back to the input
    with np.errstate(all='ignore'):
        r = ref_handler(y, w, d)
    i = len(cases)
    out['y_%d' % i], out['ref_%d' % i], out['par_%d' % i] = y, r, np.array([w, d])
    cases.append((n, w, d, kind))
    print(i, n, w, d, kind, 'nan-fallback' if np.array_equal(r, y) and n >= 10 else '')
out['n_cases'] = np.array(len(cases))
np.savez_compressed(os.path.join('tests', 'golden', 'loess_golden.npz'), **out)
