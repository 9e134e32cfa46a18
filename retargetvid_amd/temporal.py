"""Host-side temporal stages between per-frame centres and crop boxes.

These stay on the host by design (BASELINE.json north_star: "TransNetV1 shot detection and LOESS smoothing stay on
host").  Since round 4 the arithmetic is native (csrc/svc_host.cpp behind the svc_host_* entries of include/svc.h; ctypes
releases the interpreter lock for the duration of a call, which is what lets retargetvid_amd/scheduler.py run the host
stages of finished videos on a thread pool next to the feeder): one call per video for interpolation + low-pass + LOESS /
Savitzky-Golay (`centres_to_series`), one per target ratio for the boxes.  The functions below keep the signatures of the
reference's per-stage functions for callers and tests.  The Butterworth DESIGN (scipy.signal.butter / lfilter_zi) stays
SciPy's, cached per (order, cutoff): a handful of coefficients handed to the native filter.

Reference (smartVidCrop.py unless noted):
  handle_empty_centers   sc_handle_empty_centers            :1221-1300
  interpolate            interp_handler / sc_interpolate    :1528-1597
  butter_lowpass         sc_butter_lowpass_filter           :1599-1627
  loess / loess_handler  loess_handler :1629-1646 and 3rd_party_libs/loess/pyloess.py:13-95
  smoothing              sc_smoothing                       :1648-1734
  focus_stability        sc_check_for_extra_cuts + focus hold :1337-1455, :2425-2473
  shift_time             sc_shift_time                      :1740-1746
"""
import ctypes
import functools

import numpy as np

from . import _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def centres_array(dx, dy):
    """Two lists with None for a missing centre -> float64 [n, 2] with NaN."""
    return np.array([[np.nan if a is None else a, np.nan if b is None else b] for a, b in zip(dx, dy)], np.float64).reshape(-1, 2)


def fill_empty_centres(xy, segmentation_sel):
    """xy float64 [n, 2] (NaN = no centre) -> filled copy, number of centres still empty."""
    lib = _lib.load()
    cx, cy = _f64(xy[:, 0]).copy(), _f64(xy[:, 1]).copy()
    seg_sel = _i32(segmentation_sel).reshape(-1, 2)
    rc = lib.svc_host_fill_empty_centres(_p(cx), _p(cy), len(cx), _p(seg_sel), len(seg_sel))
    if rc < 0:
        raise IndexError(lib.svc_last_error().decode(errors='replace'))       # the reference's list index error
    return np.stack([cx, cy], 1), rc


def handle_empty_centers(dx, dy, segmentation_sel):
    """Fill runs of missing centres (None) from the nearest shot-consistent neighbour."""
    if not any(v is None for v in dx):
        return list(dx), list(dy)
    xy, _ = fill_empty_centres(centres_array(dx, dy), segmentation_sel)
    back = lambda col: [None if v != v else float(v) for v in col]
    return back(xy[:, 0]), back(xy[:, 1])


def interpolate(dx, dy, segmentation, segmentation_sel, true_inds):
    """Per shot: selected-frame centres -> one centre per decoded frame."""
    lib = _lib.load()
    dx, dy, ti = _f64(dx), _f64(dy), np.asarray(true_inds)
    dxi, dyi = [], []
    for seg, sel in zip(segmentation, segmentation_sel):
        si, ei = int(seg[0]), int(seg[1]) + 1
        sis, eis = int(sel[0]), int(sel[1]) + 1
        st = _f64(ti[sis:eis] - ti[sis:eis].min())
        o1, o2 = np.empty(ei - si), np.empty(ei - si)
        _lib.check(lib.svc_host_interp_segment(_p(st), _p(dx[sis:eis]), _p(dy[sis:eis]), eis - sis, ei - si, _p(o1), _p(o2)))
        dxi += o1.tolist()
        dyi += o2.tolist()
    return dxi, dyi


@functools.lru_cache(maxsize=64)
def _butter_design(order, wn):
    """(b, a, zi) of the zero-phase low-pass: SciPy's design, a few coefficients for the native filter."""
    from scipy import signal
    b, a = signal.butter(order, wn, btype='lowpass', analog=False)
    return _f64(b), _f64(a), _f64(signal.lfilter_zi(b, a))


def butter_lowpass(x, cutoff, fs, order):
    """Zero-phase Butterworth low-pass with the reference's fall-back for short series."""
    lib = _lib.load()
    b, a, zi = _butter_design(order, cutoff / (0.5 * fs))
    x = _f64(x)
    out = np.empty_like(x)
    _lib.check(lib.svc_host_lowpass(_p(b), _p(a), _p(zi), len(b), _p(x), len(x), _p(out)))
    return out


def loess(y, window, degree):
    """pyloess.Loess(arange(n), y).estimate(j, window, degree=degree) for every j."""
    lib = _lib.load()
    y = _f64(y)
    out = np.empty_like(y)
    _lib.check(lib.svc_host_loess(_p(y), len(y), int(window), int(degree), _p(out)))
    return out


def loess_handler(di, loess_filt, window, degree):
    n = len(di)
    if n < 10:
        return list(di)
    if loess_filt:
        ds = loess(di, window, degree)
        return list(di) if np.isnan(np.sum(ds)) else list(ds)
    lib = _lib.load()
    y = _f64(di)
    out = np.empty_like(y)
    if lib.svc_host_savgol(_p(y), n, int(window), int(degree), _p(out)) < 0:
        raise ValueError(lib.svc_last_error().decode(errors='replace'))
    return list(out)


def temporal_params(fr, CP):
    """-> (SvcTemporalParams, b, a, zi) for svc_host_temporal."""
    b = a = zi = None
    if CP['lp_filt']:
        b, a, zi = _butter_design(CP['lp_order'], CP['lp_cutoff'] / (0.5 * fr))
    p = _lib.SvcTemporalParams(ctypes.sizeof(_lib.SvcTemporalParams), int(bool(CP['lp_filt'])), 0 if b is None else len(b),
                               int(bool(CP['loess_filt'])), int(CP['loess_degree']), 0, float(CP['loess_w_secs']), float(fr))
    return p, b, a, zi


def centres_to_series(xy, true_inds, segmentation, segmentation_sel, fc, fr, CP):
    """sc_interpolate + sc_smoothing of one video in ONE native call: xy float64 [n_sel, 2] without NaN ->
    (xi, yi, xs, ys) float64 [fc] each (interpolated, then low-passed and LOESS / Savitzky-Golay smoothed)."""
    lib = _lib.load()
    p, b, a, zi = temporal_params(fr, CP)
    cx, cy = _f64(xy[:, 0]), _f64(xy[:, 1])
    ti, seg, sel = _i32(true_inds), _i32(segmentation).reshape(-1, 2), _i32(segmentation_sel).reshape(-1, 2)
    n_out = int((seg[:, 1] - seg[:, 0] + 1).sum())
    out = np.empty((4, max(n_out, int(fc))), np.float64)
    rc = lib.svc_host_temporal(ctypes.byref(p), _p(b) if b is not None else None, _p(a) if a is not None else None,
                               _p(zi) if zi is not None else None, _p(cx), _p(cy), len(cx), _p(ti), _p(seg), _p(sel), len(seg),
                               out.shape[1], _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]))
    _lib.check(rc)
    return out[0, :rc], out[1, :rc], out[2, :rc], out[3, :rc]


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


def boxes(dxs, dys, w_orig, h_orig, w_process, h_process, w_final, h_final, borders=(0, 0, 0, 0)):
    """sc_compute_bb's arithmetic (smartVidCrop.py:979-1048) -> (boxes int64 [fc, 4], truncated centres int64 [fc, 2],
    fbb_w, fbb_h)."""
    lib = _lib.load()
    xs, ys = _f64(dxs), _f64(dys)
    fc = len(xs)
    bb = np.empty((fc, 4), np.int64)
    ctr = np.empty((fc, 2), np.int64)
    wh = np.zeros(2, np.int32)
    brd = _i32(borders)
    _lib.check(lib.svc_host_boxes(_p(xs), _p(ys), fc, int(w_orig), int(h_orig), int(w_process), int(h_process), int(w_final),
                                  int(h_final), _p(brd), _p(bb), _p(ctr), _p(wh)))
    return bb, ctr, int(wh[0]), int(wh[1])


def focus_stability_native(xy, maps_nhw, fr, CP):
    """focus_stability on arrays, native (svc_host_focus_stability): xy float64 [n, 2]; maps_nhw: the FILTERED maps frame-major
    uint8 [n, h, w] on the host (as they come from the device: no [H,W,n] transposition).  -> (xy, jumps list, jumps_inds list);
    a jump without a statistic is the integer 255, as in the reference."""
    lib = _lib.load()
    maps = np.ascontiguousarray(maps_nhw, np.uint8)
    n, h, w = maps.shape
    cx, cy = _f64(xy[:, 0]).copy(), _f64(xy[:, 1]).copy()
    jumps = np.empty(n, np.float64)
    inds = np.empty(max(n, 1), np.int32)
    ni = _lib.check(lib.svc_host_focus_stability(_p(cx), _p(cy), n, _p(maps), h, w, float(fr), int(CP['skip']), float(CP['min_d_jump']),
                                                 float(CP['foces_stab_t']), float(CP['foces_stab_s']), _p(jumps), _p(inds)))
    jl = [255 if v == 255.0 else float(v) for v in jumps]
    return np.stack([cx, cy], 1), jl, [int(v) for v in inds[:ni]]


def focus_stability(dx, dy, smaps_hwn, fr, CP):
    """Jump statistics + focus hold (smartVidCrop.py:1337-1455, :2425-2473) with the reference's argument layout: smaps_hwn =
    the filtered maps as [H,W,n].  -> (dx, dy, jumps, jumps_inds).  (after_ingest calls focus_stability_native on the
    frame-major maps directly.)"""
    xy, jumps, inds = focus_stability_native(np.stack([_f64(dx), _f64(dy)], 1), np.transpose(np.asarray(smaps_hwn), (2, 0, 1)), fr, CP)
    return xy[:, 0].tolist(), xy[:, 1].tolist(), jumps, inds


def shift_time(bbs, shift):
    if shift > 0:
        for i in range(shift):
            bbs[-i + 1] = bbs[-1]
        for i in range(len(bbs) - shift):
            bbs[i] = bbs[i + shift]
    return bbs
