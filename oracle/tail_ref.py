"""Oracle: the per-pixel tail of the hot path in NumPy (K8, K9, K11-K14, K16, boxes).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows smartVidCrop.py:
  :1050-1059  sc_threshold
  :1062-1161  sc_clustering_filt  (COO gather in raster order, HDBSCAN, cluster weight =
              max (select_sum != 1) or sum (== 1), FIRST arg-max wins, every point whose
              label differs — noise included — is zeroed, then CLOSE 5x5; maps with
              N <= hdbscan_min+1 points, and maps where no cluster is found, are returned
              untouched and un-closed)
  :2359-2373  clustering driver loop with the cut-adjacent blend (u8 sum wraps mod 256
              before the divide)
  :1163-1219  sc_find_center_of_mass (single-cluster K-means == unweighted centroid)
  :2402-2414  centre loop (None for empty maps)
  :946-977    sc_calc_dest_size
  :979-1048   sc_compute_bb
  :927-944    bb_intersection_over_union   (== retargetvid_eval.py:10-27)

Integer/byte work throughout: the HIP path must match these bit for bit.
"""
import math

import numpy as np

from . import cv_ref, hdbscan_ref


def threshold(smaps, t):
    """smaps: u8 array, in place like the reference.  smartVidCrop.py:1057"""
    smaps[smaps < t] = 0
    return smaps


def gather_points(sal_map):
    """scipy coo_matrix(sal_map) order = np.nonzero raster order.  -> X[N,2] (row, col), W[N]."""
    rows, cols = np.nonzero(sal_map)
    return np.stack([rows, cols], axis=1).astype(np.int64), sal_map[rows, cols]


def clustering_filt(sal_map, CP, info=None, labels_fn=None):
    """smartVidCrop.py:1062-1161 on one [H,W] u8 map; returns a new array.
    labels_fn(X, min_cluster_size, min_samples) replaces the restated HDBSCAN (used by the tie-order study,
    tools/make_golden_hdbscan.py, to push another implementation's labels through the same K11-K14)."""
    sal_map = np.array(sal_map, np.uint8, copy=True)
    if np.sum(sal_map) == 0:
        return sal_map
    factor = CP['resize_factor']
    init_h, init_w = sal_map.shape
    if factor != 1.0:
        if CP['resize_type'] != 1:
            raise NotImplementedError('only resize_type 1 (bilinear) is restated')
        sal_map = cv_ref.resize_linear_factor_u8(sal_map, 1.0 / factor)
    X, W = gather_points(sal_map)
    if info is not None:
        info['n_points'] = len(X)
    if X.shape[0] > CP['hdbscan_min'] + 1:
        labels = (labels_fn or hdbscan_ref.hdbscan_labels)(X, CP['hdbscan_min'], CP['hdbscan_min_samples'])
        n_clusters = len(set(labels.tolist())) - (1 if -1 in labels else 0)
        if info is not None:
            info['labels'] = labels
        if n_clusters > 0:
            weights = []
            for i in range(n_clusters):
                sel = W[labels == i]
                weights.append(int(np.sum(sel.astype(np.int64))) if CP['select_sum'] == 1 else int(np.amax(sel)))
            max_cl = weights.index(max(weights))
            drop = labels != max_cl
            sal_map[X[drop, 0], X[drop, 1]] = 0
            if CP['op_close']:
                sal_map = cv_ref.morph_close_5x5(sal_map)
    if factor == 1.0:
        return sal_map
    return cv_ref.resize_linear_u8(sal_map, init_h, init_w)


def blend_next(cur, nxt):
    """smartVidCrop.py:2371-2373: ((next + cur) as u8, wrapping) / 2.0 -> int -> u8."""
    a = (nxt + cur).astype('float')      # u8 + u8 wraps mod 256
    a = a / 2.0
    return a.astype('int').astype(np.uint8)


def segm_cuts_of(segmentation_sel):
    cuts = [int(s[0]) for s in segmentation_sel]
    cuts.append(int(segmentation_sel[-1][1]))
    return cuts


def cluster_loop(smaps_hwn, segm_cuts, CP):
    """smartVidCrop.py:2359-2373 on [H,W,N] u8 (thresholded), in place."""
    n = smaps_hwn.shape[2]
    for i in range(n):
        smaps_hwn[:, :, i] = clustering_filt(smaps_hwn[:, :, i], CP)
        if i < n - 2:
            if any(x in segm_cuts for x in (i - 1, i, i + 1)):
                smaps_hwn[:, :, i + 1] = blend_next(smaps_hwn[:, :, i], smaps_hwn[:, :, i + 1])
    return smaps_hwn


def center_of_mass(sal_map, factor=1.0):
    """smartVidCrop.py:1163-1219 with km=True: centroid (x, y) of the non-zero pixels of the
    (nearest-down-scaled when factor != 1) map, times factor; (None, None) if empty."""
    if factor != 1.0:
        sal_map = cv_ref.resize_nearest_factor_u8(sal_map, 1.0 / factor)
    rows, cols = np.nonzero(sal_map)
    if rows.size == 0:
        return None, None
    return float(np.mean(cols.astype(np.float64))) * factor, float(np.mean(rows.astype(np.float64))) * factor


def center_argmax(sal_map):
    """sc_find_center_of_mass with km=False (smartVidCrop.py:1165-1178): the position of the first maximum in raster
    order, on the map as it is (no resize, no factor)."""
    y, x = np.unravel_index(int(np.argmax(sal_map)), sal_map.shape)
    return int(x), int(y)


def centers(smaps_hwn, CP):
    dx, dy = [], []
    for i in range(smaps_hwn.shape[2]):
        if np.sum(smaps_hwn[:, :, i]) > 0:
            if not CP.get('com_km', True):
                x, y = center_argmax(smaps_hwn[:, :, i])
            else:
                x, y = center_of_mass(smaps_hwn[:, :, i], CP['resize_factor'])
        else:
            x, y = None, None
        dx.append(x)
        dy.append(y)
    return dx, dy


def calc_dest_size(w_orig, h_orig, out_ratio):
    """smartVidCrop.py:946-977 -> (w_final, h_final, conversion_mode)."""
    a, b = (float(v) for v in out_ratio.split(':'))
    if abs(float(w_orig) / float(h_orig) - a / b) < 0.0000001:
        return w_orig, h_orig, 0
    w_f, h_f, mode = int(math.floor((a / b) * h_orig)), h_orig, 1
    if w_f > w_orig or h_f > h_orig:
        w_f, h_f, mode = w_orig, int(math.floor((b / a) * w_orig)), 2
    return w_f, h_f, mode


def compute_bb(dxs, dys, fc, w_orig, h_orig, w_process, h_process, w_final, h_final,
               borders=(0, 0, 0, 0)):
    """smartVidCrop.py:979-1048 -> (bbs[fc][4], fbb_w, fbb_h).  borders = (t, b, l, r)."""
    bt, bb, bl, br = borders
    scale_h = float(h_process) / float(h_orig)
    scale_w = float(w_process) / float(w_orig)
    xs = [int(dxs[i] / scale_w) for i in range(fc)]
    ys = [int(dys[i] / scale_h) for i in range(fc)]
    fbb_w, fbb_h = w_final, h_final
    if h_final == h_orig:
        fbb_h = h_final - bt - bb
        fbb_w = int((float(fbb_h) / float(h_final)) * w_final)
    if w_final == w_orig:
        fbb_w = w_final - bl - br
        fbb_h = int((float(fbb_w) / float(w_final)) * h_final)
    hw1 = int(fbb_w / 2.0)
    hw2 = fbb_w - hw1
    hh1 = int(fbb_h / 2.0)
    hh2 = fbb_h - hh1
    bbs = []
    for i in range(fc):
        x1, y1, x2, y2 = xs[i] - hw1, ys[i] - hh1, xs[i] + hw2, ys[i] + hh2
        if x1 < bl:
            x1 = bl
            x2 = x1 + fbb_w
        if x2 > w_orig - br:
            x2 = w_orig - br
            x1 = x2 - fbb_w
        if y1 < bt:
            y1 = bt
            y2 = y1 + fbb_h
        if y2 > h_orig - bb:
            y2 = h_orig - bb
            y1 = y2 - fbb_h
        bbs.append([x1, y1, x2, y2])
    return bbs, fbb_w, fbb_h


def iou(a, b):
    """smartVidCrop.py:927-944 / retargetvid_eval.py:10-27 (inclusive +1 pixel convention)."""
    xa, ya = max(a[0], b[0]), max(a[1], b[1])
    xb, yb = min(a[2], b[2]), min(a[3], b[3])
    inter = max(0, xb - xa + 1) * max(0, yb - ya + 1)
    area_a = (a[2] - a[0] + 1) * (a[3] - a[1] + 1)
    area_b = (b[2] - b[0] + 1) * (b[3] - b[1] + 1)
    return inter / float(area_a + area_b - inter)
