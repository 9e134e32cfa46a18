"""Host-side temporal stages between per-frame centres and crop boxes.

These stay on the host by design (BASELINE.json north_star: "TransNetV1 shot detection
and LOESS smoothing stay on host").  Same results as the reference functions, written
for speed where the reference is a Python loop (LOESS is evaluated for all frames of a
shot at once with batched 3x3 pseudo-inverses instead of one ``Loess.estimate`` call per
frame).

Reference (smartVidCrop.py unless noted):
  handle_empty_centers   sc_handle_empty_centers            :1221-1300
  interpolate            interp_handler / sc_interpolate    :1528-1597
  butter_lowpass         sc_butter_lowpass_filter           :1599-1627
  loess / loess_handler  loess_handler :1629-1646 and 3rd_party_libs/loess/pyloess.py:13-95
  smoothing              sc_smoothing                       :1648-1734
  focus_stability        sc_check_for_extra_cuts + focus hold :1337-1455, :2425-2473
  shift_time             sc_shift_time                      :1740-1746
"""
import functools

import numpy as np
from scipy import interpolate as _interp, signal as _signal


def handle_empty_centers(dx, dy, segmentation_sel):
    """Fill runs of missing centres (None) from the nearest shot-consistent neighbour."""
    dx, dy = list(dx), list(dy)
    missing = np.array([v is None for v in dx], bool)
    if not missing.any():
        return dx, dy
    starts = np.array([int(s[0]) for s in segmentation_sel])
    ends = np.array([int(s[1]) for s in segmentation_sel])
    edges = np.flatnonzero(np.diff(np.concatenate([[0], missing.view(np.int8), [0]])))
    for lo, hi in zip(edges[0::2], edges[1::2] - 1):
        d_start = int(np.min(np.abs(starts - lo)))
        d_end = int(np.min(np.abs(ends - hi)))
        src = hi + 1 if d_start < d_end else lo - 1        # a negative index wraps, as in the reference
        for j in range(lo, hi + 1):
            dx[j], dy[j] = dx[src], dy[src]
    return dx, dy


def _interp_segment(d, sampled_t, true_t):
    n = len(d)
    if n < 3:
        return [float(d[0])] * len(true_t)
    f = _interp.interp1d(sampled_t, d, fill_value='extrapolate', kind='linear' if n <= 6 else 'quadratic')
    return list(f(true_t))


def interpolate(dx, dy, segmentation, segmentation_sel, true_inds):
    """Per shot: selected-frame centres -> one centre per decoded frame."""
    dxi, dyi = [], []
    for seg, sel in zip(segmentation, segmentation_sel):
        si, ei = int(seg[0]), int(seg[1]) + 1
        sis, eis = int(sel[0]), int(sel[1]) + 1
        st = np.asarray(true_inds[sis:eis])
        st = list(st - st.min())
        tt = np.arange(0, ei - si)
        dxi += _interp_segment(dx[sis:eis], st, tt)
        dyi += _interp_segment(dy[sis:eis], st, tt)
    return dxi, dyi


@functools.lru_cache(maxsize=64)
def _butter_design(order, wn):
    return _signal.butter(order, wn, btype='lowpass', analog=False)      # the design is ~10x the cost of filtering a shot


def butter_lowpass(x, cutoff, fs, order):
    """Zero-phase Butterworth low-pass with the reference's fall-back chain for short series."""
    try:
        b, a = _butter_design(order, cutoff / (0.5 * fs))
        return _signal.filtfilt(b, a, x)
    except Exception:
        pass
    for width in (5, 3):
        try:
            y = np.convolve(x, np.ones(width), 'same') / 5
            x[2:len(x) - 2] = y[2:len(x) - 2]
            return x
        except Exception:
            pass
    return x


@functools.lru_cache(maxsize=32)
def _loess_design(n, window, degree):
    """Everything of the local regressions that does not depend on the data: neighbourhoods, tricube weights
    and, for degree > 1, the operator pinv(X^T W X) X^T W of every point (pyloess.py:60-83)."""
    with np.errstate(all='ignore'):
        nx = np.arange(n, dtype=np.float64) / (n - 1)
        h = (window - 1) // 2
        j = np.arange(n)
        # neighbourhood chosen by get_min_range (pyloess.py:27-48): the nearer of the two outer neighbours joins next,
        # the right one on a tie, clamped at the ends -> the symmetric 2h+1 points around j; an even window (never
        # produced by sc_smoothing, :1668-1670) takes one more point, decided by the same float64 comparison
        lo = np.clip(j - h, 0, n - (2 * h + 1))
        if window % 2 == 0:
            hi = lo + 2 * h
            dl = np.abs(nx[np.maximum(lo - 1, 0)] - nx[j])
            dr = np.abs(nx[np.minimum(hi + 1, n - 1)] - nx[j])
            left = (hi == n - 1) | ((lo > 0) & (dl < dr))
            left &= (j != 0)                                          # argmin at an end: arange(0, window) / arange(n - window, n)
            left |= (j == n - 1)
            lo = np.where(left, lo - 1, lo)
        idx = lo[:, None] + np.arange(window)[None, :]                # [n, window]
        dist = np.abs(nx[idx] - nx[j][:, None])
        r = dist / dist.max(axis=1, keepdims=True)
        w = np.where((r >= -1) & (r <= 1), (1.0 - np.abs(r) ** 3) ** 3, 0.0)
        if degree > 1:
            xm = nx[idx][:, :, None] ** np.arange(degree + 1)[None, None, :]     # [n, window, d+1]
            xtw = np.transpose(xm, (0, 2, 1)) * w[:, None, :]                    # X^T W
            op = np.linalg.pinv(xtw @ xm) @ xtw                                   # [n, d+1, window]
            xp = nx[j][:, None] ** np.arange(degree + 1)[None, :]
            return idx, w, nx, op, xp
        return idx, w, nx, None, None


def loess(y, window, degree):
    """pyloess.Loess(arange(n), y).estimate(j, window, degree=degree) for every j, vectorised."""
    y = np.asarray(y, np.float64)
    n = y.shape[0]
    ymin, ymax = y.min(), y.max()
    idx, w, nx, op, xp = _loess_design(n, int(window), int(degree))
    with np.errstate(all='ignore'):
        ny = (y - ymin) / (ymax - ymin)
        if degree > 1:
            beta = op @ ny[idx][:, :, None]                                       # [n, d+1, 1]
            est = np.einsum('nd,nd->n', beta[:, :, 0], xp)
        else:
            j = np.arange(n)
            sx, sy = nx[idx], ny[idx]
            sw = w.sum(axis=1)
            mx, my = (sx * w).sum(1) / sw, (sy * w).sum(1) / sw
            b = ((sx * sy * w).sum(1) - mx * my * sw) / ((sx * sx * w).sum(1) - mx * mx * sw)
            est = (my - b * mx) + b * nx[j]
        return est * (ymax - ymin) + ymin


def loess_handler(di, loess_filt, window, degree):
    n = len(di)
    if n < 10:
        return list(di)
    if loess_filt:
        ds = loess(di, window, degree)
        return list(di) if np.isnan(np.sum(ds)) else list(ds)
    return list(_signal.savgol_filter(di, window, degree))


def smoothing(dxi, dyi, segmentation, fr, CP):
    """Per shot: low-pass then LOESS (or Savitzky-Golay).  -> (dxs, dys) over all frames."""
    dxs, dys = [], []
    for seg in segmentation:
        si, ei = int(seg[0]), int(seg[1]) + 1
        cl = ei - si
        win = min(int(fr * CP['loess_w_secs']), cl - 2)
        if win % 2 == 0:
            win -= 1
        for series, out in ((dxi, dxs), (dyi, dys)):
            d = np.array(series[si:ei])
            dl = butter_lowpass(d, CP['lp_cutoff'], fr, CP['lp_order']) if CP['lp_filt'] else d
            out += loess_handler(dl, CP['loess_filt'], win, CP['loess_degree'])
    return dxs, dys


def _points_on_line(p1x, p1y, p2x, p2y, w, h, min_d):
    """get_points_on_line (smartVidCrop.py:1337-1393): integer-stepped samples strictly after p1 up
    to p2 along the dominant axis; the reference's removed ``np.int`` cast is the intended
    truncation.  -> float32 [m,2] (x, y) inside the image, or None for a jump below min_d."""
    dX, dY = p2x - p1x, p2y - p1y
    dXa, dYa = abs(dX), abs(dY)
    if dXa < min_d and dYa < min_d:
        return None
    m = int(np.ceil(max(dYa, dXa)))
    buf = np.full((m, 2), np.nan, np.float32)
    sy = np.arange(p1y - 1, p1y - dYa - 1, -1) if p1y > p2y else np.arange(p1y + 1, p1y + dYa + 1)
    sx = np.arange(p1x - 1, p1x - dXa - 1, -1) if p1x > p2x else np.arange(p1x + 1, p1x + dXa + 1)
    try:
        if p1x == p2x:
            buf[:, 0], buf[:, 1] = p1x, sy
        elif p1y == p2y:
            buf[:, 1], buf[:, 0] = p1y, sx
        elif dYa > dXa:
            buf[:, 1] = sy
            buf[:, 0] = (np.float32(dX) / np.float32(dY) * (buf[:, 1] - p1y)).astype(int) + p1x
        else:
            buf[:, 0] = sx
            buf[:, 1] = (np.float32(dY) / np.float32(dX) * (buf[:, 0] - p1x)).astype(int) + p1y
    except Exception:
        return None
    keep = (buf[:, 0] >= 0) & (buf[:, 1] >= 0) & (buf[:, 0] < w) & (buf[:, 1] < h)
    return buf[keep]


def focus_stability(dx, dy, smaps_hwn, fr, CP):
    """Jump statistics + focus hold (smartVidCrop.py:1395-1455, :2425-2473).  smaps_hwn: the
    filtered maps in the reference's [H,W,n] layout.  -> (dx, dy, jumps, jumps_inds)."""
    dx, dy = list(dx), list(dy)
    n = len(dx)
    h, w = smaps_hwn.shape[:2]
    jumps, inds = [255] * n, []
    for i in range(1, n):
        pts = _points_on_line(dx[i - 1], dy[i - 1], dx[i], dy[i], w, h, CP['min_d_jump'])
        if pts is not None:
            ok = ~np.isnan(pts[:, 0])
            if ok.any():
                xs = np.floor(pts[ok, 0]).astype(np.int64)
                ys = np.floor(pts[ok, 1]).astype(np.int64)
                jumps[i] = float(smaps_hwn[ys, xs, i].astype(np.float64).sum()) / float(ok.sum())
        if jumps[i] < CP['foces_stab_t']:
            inds.append(i)
    for a, b in zip(inds[:-1], inds[1:]):
        start, end = max(a - 1, 0), min(b + 1, n - 1)
        if ((end - start) * CP['skip']) / fr <= CP['foces_stab_s']:
            for j in range(end - start):
                dx[start + j], dy[start + j] = dx[start], dy[start]
    return dx, dy, jumps, inds


def shift_time(bbs, shift):
    if shift > 0:
        for i in range(shift):
            bbs[-i + 1] = bbs[-1]
        for i in range(len(bbs) - shift):
            bbs[i] = bbs[i + shift]
    return bbs
