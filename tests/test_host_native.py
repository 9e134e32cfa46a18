"""The native host stages (csrc/svc_host.cpp, svc_host_* in include/svc.h) against the oracle's restatement of the
reference's temporal functions (oracle/temporal_ref.py: SciPy / pyloess driven exactly as smartVidCrop.py:1221-1300,
:1528-1734 drives them) and against SciPy itself.  No GPU: the entries take host pointers only.

Bars: bit for bit where the native code performs SciPy's own operations in SciPy's order (linear interpolation, the
filtfilt chain, the moving-average fall-back, the empty-centre fill, the box arithmetic); 1e-9 px for the quadratic
spline (another banded LU than LAPACK's), 1e-6 px for LOESS at the degrees in use (the reference's own pseudo-inverse
route is only that accurate: tests/test_oracle_loess.py) and 1e-9 px for Savitzky-Golay."""
import numpy as np
import pytest
from scipy import interpolate as sp_interp, signal as sp_signal

from oracle import pipeline_ref as P, tail_ref as T, temporal_ref as TR
from retargetvid_amd import smartVidCrop as S, temporal


def test_linear_interpolation_is_bit_for_bit_scipy():
    rng = np.random.RandomState(0)
    for n in (3, 4, 5, 6):
        for trial in range(20):
            st = np.sort(rng.choice(np.arange(1, 40), n - 1, replace=False))
            st = np.concatenate([[0], st]).astype(np.float64)
            d1, d2 = rng.uniform(0, 250, n), rng.uniform(0, 140, n)
            n_out = int(st[-1]) + rng.randint(1, 8)                      # a few frames behind the last sample: extrapolation
            o1, o2 = np.empty(n_out), np.empty(n_out)
            lib = temporal._lib.load()
            assert lib.svc_host_interp_segment(temporal._p(st), temporal._p(d1), temporal._p(d2), n, n_out, temporal._p(o1), temporal._p(o2)) == 0
            for d, o in ((d1, o1), (d2, o2)):
                ref = sp_interp.interp1d(st, d, fill_value='extrapolate', kind='linear')(np.arange(n_out))
                assert np.array_equal(o, ref)


def test_quadratic_spline_matches_scipy():
    rng = np.random.RandomState(1)
    worst = 0.0
    for n in (7, 8, 9, 20, 57, 214):
        for trial in range(6):
            gaps = rng.choice([1, 2, 6, 6, 6, 6], n - 1)                # the selected frames: mostly `skip` apart, denser behind a cut
            st = np.concatenate([[0], np.cumsum(gaps)]).astype(np.float64)
            d1, d2 = np.cumsum(rng.randn(n)) * 4 + 120, rng.uniform(0, 140, n)
            n_out = int(st[-1]) + 6
            o1, o2 = np.empty(n_out), np.empty(n_out)
            lib = temporal._lib.load()
            assert lib.svc_host_interp_segment(temporal._p(st), temporal._p(d1), temporal._p(d2), n, n_out, temporal._p(o1), temporal._p(o2)) == 0
            for d, o in ((d1, o1), (d2, o2)):
                ref = np.asarray(TR.interp_segment(list(d), list(st.astype(int)), np.arange(n_out)))
                worst = max(worst, float(np.abs(o - ref).max()))
    assert worst < 1e-9, worst
    # one or two samples: the first value everywhere (interp_handler's l < 3 rule)
    o1, o2 = np.empty(5), np.empty(5)
    st, d = np.array([0.0, 3.0]), np.array([7.5, 9.0])
    assert temporal._lib.load().svc_host_interp_segment(temporal._p(st), temporal._p(d), temporal._p(d), 2, 5, temporal._p(o1), temporal._p(o2)) == 0
    assert (o1 == 7.5).all()
    with pytest.raises(Exception):
        temporal.interpolate([1.0] * 8, [1.0] * 8, [[0, 20]], [[0, 7]], [0, 3, 3, 9, 12, 15, 18, 20])      # duplicate sample times


def test_lowpass_is_bit_for_bit_scipy_filtfilt_and_the_short_series_fallback():
    rng = np.random.RandomState(2)
    for order, cutoff, fs in ((5, 2, 30.0), (2, 1, 30.0), (5, 2, 25.0), (3, 2.5, 29.97)):
        b, a = sp_signal.butter(order, cutoff / (0.5 * fs), btype='lowpass', analog=False)
        edge = 3 * (order + 1)
        for n in (edge + 1, edge + 2, 40, 131, 700):
            x = np.cumsum(rng.randn(n)) * 3 + 100
            assert np.array_equal(temporal.butter_lowpass(x.copy(), cutoff, fs, order), sp_signal.filtfilt(b, a, x))
        for n in (1, 2, 4, 5, 6, 9, edge):                                 # filtfilt raises there: the 5-point moving average
            x = np.cumsum(rng.randn(n)) * 3 + 100
            assert np.array_equal(temporal.butter_lowpass(x.copy(), cutoff, fs, order), TR.butter_lowpass(x.copy(), cutoff, fs, order))


def test_loess_and_savgol_match_the_oracle():
    rng = np.random.RandomState(3)
    worst = {1: 0.0, 2: 0.0}
    for n, w in ((10, 7), (30, 11), (61, 59), (200, 59), (643, 59), (40, 12), (35, 34)):
        y = np.cumsum(rng.randn(n)) * 3 + 120
        for deg in (1, 2):
            ref = np.array([TR.loess_estimate(np.arange(n), y, j, w, deg) for j in range(n)])
            worst[deg] = max(worst[deg], float(np.abs(temporal.loess(y, w, deg) - ref).max()))
    assert worst[1] < 1e-9 and worst[2] < 1e-6, worst
    assert np.isnan(temporal.loess(np.full(30, 4.25), 11, 2)).all()       # constant series: 0 / 0 in the reference
    assert temporal.loess_handler(np.full(30, 4.25), 1, 11, 2) == [4.25] * 30
    assert temporal.loess_handler(np.arange(9.0), 1, 7, 2) == list(np.arange(9.0))        # < 10 frames: untouched
    for n, w, deg in ((10, 7, 2), (30, 11, 2), (200, 59, 2), (200, 59, 3), (15, 13, 2), (64, 59, 1)):
        y = np.cumsum(rng.randn(n)) * 3 + 120
        ref = sp_signal.savgol_filter(y, w, deg)
        assert np.abs(np.array(temporal.loess_handler(y, 0, w, deg)) - ref).max() < 1e-9
    with pytest.raises(ValueError):
        temporal.loess_handler(np.arange(12.0), 0, 1, 2)                   # window <= degree: savgol_filter raises too


def _case(rng, n, trans):
    true_inds, m2o, _ = P.select_frames(n, n, trans, 6, 2000)
    seg = P.scenes_from_trans_inds(trans, n)
    seg_sel = np.array([[m2o[v] for v in r] for r in seg])
    k = len(true_inds)
    dx = list(np.cumsum(rng.randn(k)) * 3 + 120)
    dy = list(np.cumsum(rng.randn(k)) * 2 + 70)
    return true_inds, seg, seg_sel, dx, dy


def test_one_call_per_video_equals_the_stage_functions_and_the_oracle():
    rng = np.random.RandomState(4)
    for n, trans in ((200, [0, 120, 128, 200]), (61, [0, 61]), (452, [0, 9, 230, 452]), (90, [0, 4, 86, 90])):
        true_inds, seg, seg_sel, dx, dy = _case(rng, n, trans)
        for cp in (P.init_crop_params(), P.init_crop_params(use_best_settings=True), dict(P.init_crop_params(), loess_degree=1),
                   dict(P.init_crop_params(), lp_filt=0)):
            xy = np.stack([dx, dy], 1)
            xi, yi, xs, ys = temporal.centres_to_series(xy, true_inds, seg, seg_sel, n, 30.0, cp)
            a = temporal.interpolate(dx, dy, seg, seg_sel, true_inds)
            sa = temporal.smoothing(a[0], a[1], seg, 30.0, cp)
            assert np.array_equal(xi, a[0]) and np.array_equal(yi, a[1]) and len(xi) == n       # one code path, two doors
            assert np.array_equal(xs, sa[0]) and np.array_equal(ys, sa[1])
            b = TR.interpolate_centres(dx, dy, seg, seg_sel, true_inds)
            sb = TR.smoothing(b[0], b[1], seg, 30.0, cp)
            assert np.abs(xi - b[0]).max() < 1e-9 and np.abs(yi - b[1]).max() < 1e-9
            assert np.abs(xs - sb[0]).max() < 1e-6 and np.abs(ys - sb[1]).max() < 1e-6


def test_empty_centre_fill_and_boxes_are_the_oracles():
    rng = np.random.RandomState(5)
    for trial in range(40):
        n = rng.randint(6, 40)
        cuts = sorted(set([0] + list(rng.randint(1, n - 1, rng.randint(0, 3)))))
        seg_sel = np.array([[cuts[i], (cuts[i + 1] - 1 if i + 1 < len(cuts) else n - 1)] for i in range(len(cuts))])
        dx = [None if rng.rand() < 0.3 else float(rng.uniform(0, 250)) for _ in range(n)]
        dx[rng.randint(0, n)] = 3.0
        dy = [None if v is None else v * 0.5 for v in dx]
        try:
            ref = TR.handle_empty_centers(list(dx), list(dy), seg_sel)
        except IndexError:
            with pytest.raises(IndexError):
                temporal.handle_empty_centers(dx, dy, seg_sel)
            continue
        assert temporal.handle_empty_centers(dx, dy, seg_sel) == ref
    for ratio in ('1:3', '3:1', '4:5', '16:9', '1:1'):
        for (w, h, wp, hp) in ((640, 360, 250, 140), (480, 640, 187, 250), (1920, 1080, 250, 140)):
            wf, hf, _ = T.calc_dest_size(w, h, ratio)
            xs, ys = rng.uniform(-5, wp + 5, 300), rng.uniform(-5, hp + 5, 300)
            ref = T.compute_bb(list(xs), list(ys), 300, w, h, wp, hp, wf, hf)
            bb, ctr, fw, fh = temporal.boxes(xs, ys, w, h, wp, hp, wf, hf)
            assert bb.tolist() == ref[0] and (fw, fh) == ref[1:]


def test_focus_stability_is_the_oracles_to_the_last_bit():
    """svc_host_focus_stability against oracle/temporal_ref.focus_stability (get_points_on_line + sc_check_for_extra_cuts + the
    hold loop, smartVidCrop.py:1337-1455, :2425-2473): the float32 buffer / float32 slope arithmetic, NumPy's arange lengths,
    the sample-count mismatch that makes NumPy raise (-> no statistic), moves along one axis, integer-valued centres, centres
    outside the image.  Jump statistics, held centres and the list of low-saliency jumps: equal, not close."""
    rng = np.random.RandomState(0)
    n_stats = 0
    for trial in range(120):
        n = rng.randint(7, 40)
        h, w = (35, 62) if trial % 2 else (140, 250)
        maps = np.zeros((n, h, w), np.uint8)
        for i in range(n):
            maps[i] = np.where(rng.rand(h, w) < rng.choice([0.02, 0.2, 0.6]), rng.randint(90, 256, (h, w)), 0)
        kind = trial % 4
        if kind == 0:
            dx, dy = rng.uniform(-3, w + 3, n), rng.uniform(-3, h + 3, n)
        elif kind == 1:
            dx, dy = np.round(rng.uniform(0, w, n)), np.round(rng.uniform(0, h, n))
            dx[3], dy[5] = dx[2], dy[4]
        elif kind == 2:
            dx, dy = np.cumsum(rng.randn(n) * 0.7) + w / 2, np.cumsum(rng.randn(n) * 0.4) + h / 2
        else:
            dx, dy = rng.uniform(0, w, n), np.full(n, float(rng.randint(0, h)))
        dx, dy = [float(v) for v in dx], [float(v) for v in dy]          # Python floats, as the pipeline hands them on
        hwn = np.ascontiguousarray(np.transpose(maps, (1, 2, 0)))
        for best in (True, False):
            CP = dict(P.init_crop_params(best), foces_stab_t=int(rng.choice([60, 120, 200])))
            ref = TR.focus_stability(list(dx), list(dy), hwn, 30.0, CP)
            got = temporal.focus_stability(list(dx), list(dy), hwn, 30.0, CP)
            assert got[0] == list(ref[0]) and got[1] == list(ref[1]) and got[2] == list(ref[2]) and got[3] == list(ref[3]), (trial, best)
            n_stats += sum(1 for v in got[2] if v != 255)
    assert n_stats > 2000
