"""tests/golden/{npsort_golden,hdbscan_sklearn,hdbscan_tieorder}.npz -- the HDBSCAN slice of the oracle, pinned.

The script re-runs itself with numpy's SIMD sort dispatch disabled (NPY_DISABLE_CPU_FEATURES): numpy's default
argsort then takes the scalar introsort that every CPU runs under the reference's pinned numpy 1.19
(README.md:83-92: scipy 1.5.1 / scikit-learn 0.24.1 / torch 1.7.1; the AVX-512 / AVX2 argsort arrived in numpy 1.25),
and scikit-learn's HDBSCAN (a port of hdbscan 0.8.x; `_process_mst` sorts the MST edges with np.argsort exactly like
hdbscan_.py `_hdbscan_generic`, call site smartVidCrop.py:1099) produces what the reference's stack produces.

Part 0  npsort_golden.npz     numpy's own permutations (scalar path) of tie-heavy, sorted, reversed, random and
                              adversarial (depth limit -> heapsort) key arrays: pins oracle/npsort_ref.py.
Part 1  hdbscan_sklearn.npz   sklearn labels on 12 seeded pixel sets with the min_samples+1 mapping;
                              oracle/hdbscan_ref.py with order='numpy' must reproduce ALL of them bit for bit;
                              `stable_i` says whether the device path's stable order gives the same labels.
Part 2  hdbscan_tieorder.npz  what the tie order does downstream: thresholded saliency maps of the benchmark workload
                              (synth.blob_frames -> oracle UNISAL forward -> threshold; both published parameter sets)
                              clustered in (a) stable order (device convention) and (b) numpy order (the reference;
                              checked against sklearn on every map), each pushed through the same K11-K14 (cluster
                              choice, filter, CLOSE, centre; smartVidCrop.py:1099-1128, :1163-1219) and the box
                              arithmetic (:979-1048); then whole multi-shot videos through the oracle pipeline in both
                              orders: final crop windows and IoU against fixed synthetic annotations.

Run from the repo root:  python tools/make_golden_hdbscan.py"""
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
from oracle import cv_ref, hdbscan_ref as H, npsort_ref, pipeline_ref as P, tail_ref as T, unisal_ref as U     # noqa: E402
from retargetvid_amd import synth, weights                                                                     # noqa: E402

warnings.filterwarnings('ignore')


# ---- part 0: the sort ------------------------------------------------------------------------------------
def killer(n):
    """McIlroy's adversary run against the restated introsort: a key array on which it degenerates, so that the
    depth limit is hit and popped ranges are heap-sorted."""
    gas = n
    val = [gas] * n
    state = dict(nsolid=0, cand=0)

    class Item:
        __slots__ = ('i',)

        def __init__(self, i):
            self.i = i

        def __lt__(self, other):
            x, y = self.i, other.i
            if val[x] == gas and val[y] == gas:
                f = x if x == state['cand'] else y
                val[f] = state['nsolid']
                state['nsolid'] += 1
            if val[x] == gas:
                state['cand'] = x
            elif val[y] == gas:
                state['cand'] = y
            return val[x] < val[y]

    npsort_ref.argsort([Item(i) for i in range(n)])
    return np.array([v if v != gas else n for v in val], np.float64)


def part0():
    r = np.random.RandomState(1)
    out, n_heap = {}, 0
    arrays = []
    for t in range(60):
        n = int(r.choice([1, 2, 3, 16, 17, 18, 33, 100, 257, 1000, 1642, 5000]))
        kind = t % 5
        w = [r.randint(1, 6, n), r.randint(1, 40, n), np.sort(r.randint(1, 40, n)), np.sort(r.randint(1, 400, n))[::-1].copy(),
             r.randint(1, 100000, n)][kind]
        arrays.append(w.astype(np.float64))
    arrays += [killer(n) for n in (64, 300, 2000)]
    for i, w in enumerate(arrays):
        o = np.argsort(w)
        st = {}
        mine = np.asarray(npsort_ref.argsort(w.tolist(), st))
        assert np.array_equal(o, mine), 'npsort restatement differs from numpy on array %d' % i
        n_heap += st.get('heapsorts', 0)
        out['w_%d' % i], out['o_%d' % i] = w.astype(np.int32), o.astype(np.int32)
    assert n_heap > 0, 'no array reached the heapsort fall-back'
    out['n'] = np.array(len(arrays))
    out['numpy_version'] = np.array(np.__version__)
    np.savez_compressed('tests/golden/npsort_golden.npz', **out)
    print('part 0: %d arrays identical to numpy %s (scalar path), %d heapsort fall-backs' % (len(arrays), np.__version__, n_heap))


# ---- part 1 ------------------------------------------------------------------------------------------------
def sk_labels(X, mcs, ms):
    k = H.effective_min_samples(len(X), mcs, ms)
    return HDBSCAN(min_cluster_size=mcs, min_samples=k + 1, metric='sqeuclidean', allow_single_cluster=True,
                   algorithm='brute').fit_predict(np.asarray(X, float)).astype(np.int64)


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


def part1():
    out = {}
    cases = [(0, 1, 26, 0), (1, 2, 26, 0), (2, 3, 26, 0), (3, 4, 26, 0), (4, 2, 5, 3), (5, 3, 5, 3), (6, 5, 26, 0),
             (7, 1, 5, 3), (8, 3, 10, 0), (9, 6, 15, 4), (10, 2, 26, 0), (11, 4, 5, 3)]
    for i, (seed, nb, mcs, ms) in enumerate(cases):
        X = blob_points(seed, nb, noise=[0.002, 0.01, 0.0][seed % 3])
        sk = sk_labels(X, mcs, ms or None)
        ref, tr = H.hdbscan_labels(X, mcs, ms or None, return_tree=True, order='numpy')
        assert np.array_equal(ref, sk), 'case %d: the restatement in numpy order differs from sklearn' % i
        stable = H.hdbscan_labels(X, mcs, ms or None, order='stable')
        out['X_%d' % i] = X.astype(np.int16)
        out['params_%d' % i] = np.array([mcs, ms])
        out['labels_%d' % i] = sk.astype(np.int32)
        out['order_%d' % i] = np.argsort(tr['mst'][2].astype(np.float64)).astype(np.int32)      # numpy's own permutation
        out['stable_%d' % i] = np.array(bool((stable == sk).all()))
        print('part 1: case %2d  N=%4d  numpy order == sklearn; stable order %s' %
              (i, len(X), 'identical' if (stable == sk).all() else 'agrees on %.3f of the points' % (stable == sk).mean()))
    out['n_cases'] = np.array(len(cases))
    out['numpy_version'] = np.array(np.__version__)
    np.savez_compressed('tests/golden/hdbscan_sklearn.npz', **out)


# ---- part 2 ------------------------------------------------------------------------------------------------
def box_1to3(cx, cy):
    bbs, _, _ = T.compute_bb([cx], [cy], 1, 640, 360, 250, 140, 120, 360)
    return bbs[0]


def part2():
    import torch
    torch.set_num_threads(8)
    sd = weights.make_synthetic_state_dict(0)
    out, rows = {}, []
    idx = 0
    for name, CP in (('default', P.init_crop_params()), ('best', P.init_crop_params(True))):
        for seed in range(7):
            frames = synth.blob_frames(36, 360, 640, seed=300 + seed)[::4]           # 9 frames of a moving-blob video
            small = np.stack([cv_ref.resize_linear_u8(f, 140, 250) for f in frames])
            maps = U.saliency_u8(sd, small)                                          # [140,250,n]
            T.threshold(maps, CP['t_threshold'])
            for j in range(maps.shape[2]):
                m = np.ascontiguousarray(maps[:, :, j])
                ia, ib, ic = {}, {}, {}
                fa = T.clustering_filt(m, CP, ia, labels_fn=lambda X, a, b: H.hdbscan_labels(X, a, b, order='stable'))
                fb = T.clustering_filt(m, CP, ib, labels_fn=lambda X, a, b: H.hdbscan_labels(X, a, b, order='numpy'))
                T.clustering_filt(m, CP, ic, labels_fn=sk_labels)
                la, lb, lc = ia.get('labels'), ib.get('labels'), ic.get('labels')
                assert (lb is None and lc is None) or np.array_equal(lb, lc), 'numpy-order restatement differs from sklearn'
                ca = T.center_of_mass(fa, CP['resize_factor']) if fa.any() else (None, None)
                cb = T.center_of_mass(fb, CP['resize_factor']) if fb.any() else (None, None)
                lab_diff = int((la != lb).sum()) if la is not None else 0
                dc = 0.0 if ca[0] is None else float(np.hypot(ca[0] - cb[0], ca[1] - cb[1]))
                ba, bb = (box_1to3(*ca), box_1to3(*cb)) if ca[0] is not None else ([0] * 4, [0] * 4)
                rows.append(dict(set=name, seed=seed, frame=j, n_points=int(ia.get('n_points', 0)), labels_differ=lab_diff,
                                 filtered_pixels_differ=int((fa != fb).sum()), d_centre_px=dc,
                                 d_box_px=int(np.abs(np.array(ba) - np.array(bb)).max())))
                out['map_%d' % idx] = np.packbits(m > 0)                             # the point set ...
                out['val_%d' % idx] = m[m > 0]                                       # ... and its values, raster order
                out['sk_%d' % idx] = (lc if lc is not None else np.zeros(0)).astype(np.int16)
                out['cen_%d' % idx] = np.array([np.nan if v is None else v for v in (ca + cb)])
                out['set_%d' % idx] = np.array(name == 'best')
                idx += 1
    per = {}
    for s in ('default', 'best'):
        rs = [r for r in rows if r['set'] == s]
        per[s] = dict(maps=len(rs), label_diff=sum(r['labels_differ'] > 0 for r in rs),
                      filtered_map_diff=sum(r['filtered_pixels_differ'] > 0 for r in rs),
                      centre_diff=sum(r['d_centre_px'] > 0 for r in rs), max_d_centre_px=max(r['d_centre_px'] for r in rs),
                      max_d_box_px=max(r['d_box_px'] for r in rs), mean_points=float(np.mean([r['n_points'] for r in rs])))
    # whole videos: final crop windows and IoU against fixed synthetic annotations, both orders
    vids = []
    for k in range(10):
        n = 60 + 6 * k
        video = dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.blob_frames(n, 360, 640, seed=400 + k),
                     trans_inds=[0, 25 + k, n])
        rng = np.random.RandomState(900 + k)
        gx = np.clip(np.cumsum(rng.randn(n) * 3) + rng.randint(100, 420), 0, 520).astype(int)
        gt = np.stack([gx, np.zeros(n, int), gx + 120, np.full(n, 360)], 1)
        res = {}
        for order in ('stable', 'numpy'):
            H.DEFAULT_ORDER = order
            VD = P.smart_vid_crop(dict(video), dict(P.init_crop_params(), out_ratio='1:3'), sd)
            res[order] = np.array(VD['bbs'])
        H.DEFAULT_ORDER = 'stable'
        iou = {o: float(np.mean([T.iou(a, b) for a, b in zip(gt.tolist(), res[o].tolist())])) for o in res}
        d = np.abs(res['stable'] - res['numpy'])
        vids.append(dict(video=k, frames=n, frames_with_different_box=int((d.max(1) > 0).sum()), max_d_box_px=int(d.max()),
                         mean_iou_stable=iou['stable'], mean_iou_numpy=iou['numpy'], d_mean_iou=abs(iou['stable'] - iou['numpy'])))
        print('part 2: video %d  %3d frames  boxes differ on %3d, max %d px, |d mean IoU| %.2e' %
              (k, n, vids[-1]['frames_with_different_box'], vids[-1]['max_d_box_px'], vids[-1]['d_mean_iou']))
    summ = dict(numpy=np.__version__, per_set=per, videos=vids,
                videos_total=dict(frames=sum(v['frames'] for v in vids), frames_with_different_box=sum(v['frames_with_different_box'] for v in vids),
                                  max_d_box_px=max(v['max_d_box_px'] for v in vids), max_d_mean_iou=max(v['d_mean_iou'] for v in vids)))
    print(json.dumps(dict(per_set=per, videos_total=summ['videos_total']), indent=1))
    out['n_maps'] = np.array(idx)
    out['summary'] = np.array(json.dumps(summ))
    out['rows'] = np.array(json.dumps(rows))
    np.savez_compressed('tests/golden/hdbscan_tieorder.npz', **out)


if __name__ == '__main__':
    part0()
    part1()
    part2()
