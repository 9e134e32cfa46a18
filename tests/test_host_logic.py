"""Host-side product code (no GPU): parameters, frame bookkeeping, temporal stages, boxes,
BN folding, blob packing, result files — each against the oracle's literal restatement."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pipeline_ref as P, tail_ref as T, temporal_ref as TR, unisal_ref as U
from retargetvid_amd import smartVidCrop as S, temporal, weights


def test_crop_params_match_reference_values():
    for best in (False, True):
        a, b = S.sc_init_crop_params(use_best_settings=best), P.init_crop_params(best)
        assert a == b and len(a) == 31
    d = S.sc_init_crop_params()
    assert (d['t_threshold'], d['hdbscan_min'], d['hdbscan_min_samples'], d['select_sum'], d['skip']) == (120, 26, None, 2, 6)
    assert S.smart_crop_version() == '1.4.0' and S.smartVidCrop is S.smart_vid_crop


def test_frame_selection_matches_oracle():
    for n, trans, rb in [(450, [0, 450], 2000), (90, [0, 40, 90], 2000), (300, [0, 100, 101, 250, 300], 120), (7, [0, 7], 2000)]:
        a = S._select_frames(n, n, trans, 6, rb)
        b = P.select_frames(n, n, trans, 6, rb)
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]


def test_blend_flags_match_reference_loop():
    seg_sel = np.array([[0, 7], [7, 16]])
    flags = S.blend_flags(17, seg_sel)
    cuts = T.segm_cuts_of(seg_sel)
    ref = [int(i < 17 - 2 and any(x in cuts for x in (i - 1, i, i + 1))) for i in range(17)]
    assert flags.tolist() == ref and flags[0] == 1 and flags[1] == 1 and flags[2] == 0


def test_empty_centre_fill():
    seg_sel = np.array([[0, 5], [6, 11]])
    dx = [None, 1.0, 2.0, None, None, 5.0, None, 7.0, 8.0, 9.0, 10.0, None]
    dy = [None if v is None else v * 2 for v in dx]
    assert temporal.handle_empty_centers(dx, dy, seg_sel) == TR.handle_empty_centers(dx, dy, seg_sel)
    full = [1.0] * 12
    assert temporal.handle_empty_centers(full, full, seg_sel) == (full, full)


def test_interpolation_lowpass_loess_match_oracle():
    rng = np.random.RandomState(0)
    CP = P.init_crop_params()
    true_inds, m2o, _ = P.select_frames(200, 200, [0, 120, 128, 200], 6, 2000)
    seg = P.scenes_from_trans_inds([0, 120, 128, 200], 200)
    seg_sel = np.array([[m2o[v] for v in r] for r in seg])
    dx = list(np.cumsum(rng.randn(len(true_inds))) * 3 + 120)
    dy = list(np.cumsum(rng.randn(len(true_inds))) * 2 + 70)
    a = temporal.interpolate(dx, dy, seg, seg_sel, true_inds)
    b = TR.interpolate_centres(dx, dy, seg, seg_sel, true_inds)
    assert np.allclose(a[0], b[0], rtol=0, atol=1e-12) and np.allclose(a[1], b[1], rtol=0, atol=1e-12) and len(a[0]) == 200
    for cp in (CP, dict(CP, loess_filt=0), dict(CP, loess_degree=1), dict(CP, lp_filt=0)):
        sa = temporal.smoothing(a[0], a[1], seg, 30.0, cp)
        sb = TR.smoothing(b[0], b[1], seg, 30.0, cp)
        assert np.allclose(sa[0], sb[0], rtol=0, atol=1e-7) and np.allclose(sa[1], sb[1], rtol=0, atol=1e-7)
    y = rng.randn(60).cumsum()
    ref = [TR.loess_estimate(np.arange(60), y, j, 21, 2) for j in range(60)]
    assert np.allclose(temporal.loess(y, 21, 2), ref, rtol=0, atol=1e-9)
    assert np.isnan(temporal.loess(np.ones(30), 11, 2)).all()          # constant series -> NaN -> handler keeps input
    assert temporal.loess_handler(np.ones(30), 1, 11, 2) == [1.0] * 30


def test_boxes_match_oracle():
    rng = np.random.RandomState(1)
    for ratio in ('1:3', '3:1', '4:5', '16:9'):
        VD = dict(w_orig=640, h_orig=360, h_process=140, w_process=250, fc=50)
        S.sc_calc_dest_size(VD, {'out_ratio': ratio})
        assert (VD['w_final'], VD['h_final'], VD['conversion_mode']) == T.calc_dest_size(640, 360, ratio)
        xs, ys = list(rng.uniform(-5, 255, 50)), list(rng.uniform(-5, 145, 50))
        ref = T.compute_bb(list(xs), list(ys), 50, 640, 360, 250, 140, VD['w_final'], VD['h_final'])
        VD['dxs'], VD['dys'] = list(xs), list(ys)
        S.sc_compute_bb(VD, {})
        assert VD['bbs'] == ref[0] and (VD['fbb_w'], VD['fbb_h']) == ref[1:]


def test_bn_folding_equals_unfolded_block(synthetic_sd):
    """fold_state_dict (product) == conv -> eval BN (oracle formulation) on one inverted residual."""
    layers = {l['name']: l for l in weights.fold_state_dict(synthetic_sd)}
    x = torch.randn(1, 24, 9, 11)
    P_ = U._SD(synthetic_sd)
    ref = U._inverted_residual(P_, x, 'cnn.features.3.conv', 24, 24, 1, 6, True)
    e, d, p = layers['f3.expand'], layers['f3.dw'], layers['f3.project']
    y = torch.clamp(F.conv2d(x, torch.from_numpy(e['w']).reshape(144, 24, 1, 1), torch.from_numpy(e['b'])), 0, 6)
    y = torch.clamp(F.conv2d(y, torch.from_numpy(d['w'].T.copy()).reshape(144, 1, 3, 3), torch.from_numpy(d['b']),
                             1, 1, 1, 144), 0, 6)
    y = F.conv2d(y, torch.from_numpy(p['w']).reshape(24, 144, 1, 1), torch.from_numpy(p['b'])) + x
    assert (y - ref).abs().max() < 1e-4
    assert layers['stem']['w'].shape == (3, 3, 3, 32) and layers['smooth_phase']['w'].shape == (8, 8, 7, 7)
    k = synthetic_sd['smoothing_salicon.weight'].reshape(41, 41).astype(np.float64)
    assert np.allclose(layers['smooth_phase']['w'].reshape(64, 49).sum(1), k.sum(), atol=1e-6)


def test_smoothing_phase_table_equals_upsample_pad_conv(synthetic_sd):
    k = torch.from_numpy(synthetic_sd['smoothing_salicon.weight'])
    low = torch.randn(1, 1, 6, 9)
    ref = F.conv2d(F.pad(F.interpolate(low, size=(48, 72), mode='nearest'), [20] * 4, mode='replicate'), k)[0, 0]
    tab = weights.smoothing_phase_table(synthetic_sd['smoothing_salicon.weight'])
    L = low[0, 0].numpy()
    out = np.zeros((48, 72))
    for y in range(48):
        for x in range(72):
            ys = np.clip(y // 8 + np.arange(-3, 4), 0, 5)
            xs = np.clip(x // 8 + np.arange(-3, 4), 0, 8)
            out[y, x] = (tab[y % 8, x % 8] * L[np.ix_(ys, xs)]).sum()
    assert np.abs(out - ref.numpy()).max() < 1e-5


def test_blob_layout(synthetic_sd):
    layers = weights.fold_state_dict(synthetic_sd)
    blob = weights.pack_blob(layers)
    magic, nt = np.frombuffer(blob[:16], np.uint64)
    assert magic == weights.BLOB_MAGIC and nt == sum(2 if 'b' in l else 1 for l in layers)
    table = np.frombuffer(blob[16:16 + 16 * int(nt)], np.uint64).reshape(-1, 2)
    first = np.frombuffer(blob, np.float32, count=int(table[0, 1]), offset=int(table[0, 0]) * 4)
    assert np.array_equal(first, layers[0]['w'].ravel()) and int(table[0, 1]) == 864
    assert all(int(o) % 16 == 0 for o in table[:, 0])


def test_result_files_roundtrip_through_evaluator_parser(tmp_path):
    from retargetvid_amd import evaluate as E
    VD = {'bbs': [[0, 0, 120, 360], [5, 0, 125, 360]]}
    info = {'cuts_clust': 0, 't__clustering': '  0.010s,  0.100%', 't_total': '  0.050s,  0.500%'}
    p = S.write_results(str(tmp_path / 'run_a'), '001', '1:3', VD, info)
    assert os.path.basename(p) == '001_1-3.txt' and open(p).read() == '0,0,120,360\n5,0,125,360\n'
    assert E._read_boxes(open(p).read()).tolist() == VD['bbs']
    stats = E.parse_info_stats({'1-3': {1: open(str(tmp_path / 'run_a' / '001_1-3_info.txt')).read()}, '3-1': {}})
    assert stats['1-3']['t__clustering'] == [0.1] and stats['1-3']['t_total'] == [0.5] and stats['1-3']['cuts_clust'] == [0]


def test_evaluator_short_run_file_and_extra_cut_lines():
    """retargetvid_eval.py:163-178 scores the rows a short file has (print + break); :208-218 parses the
    'cuts_extra:' / 'no_extra_cuts:' info lines into the ecm / eca columns."""
    import warnings
    from oracle import tail_ref as T
    from retargetvid_amd import evaluate as E
    v = E.VID_INDS[0]
    gt = np.array([[0, 0, 120, 360]] * 5, np.int32)
    annots = [{ar: {v: gt.copy()} for ar in E.ARS} for _ in range(6)]
    boxes = {'1-3': {v: np.array([[10, 0, 130, 360]] * 3, np.int32)}, '3-1': {}}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        g, m, index = E.pair_boxes(annots, boxes)
    assert len(w) == 1 and 'scoring the first 3' in str(w[0].message)
    assert g.shape == m.shape == (18, 4) and all(n == 3 for *_, n in index)
    ious = np.array([T.iou(a, b) for a, b in zip(g.tolist(), m.tolist())])
    w_, b_, mean = E.aggregate(ious, index)['1-3']
    assert abs(mean - 100 * T.iou([0, 0, 120, 360], [10, 0, 130, 360])) < 1e-9
    import pytest
    with pytest.raises(ValueError):
        E.pair_boxes(annots, {'1-3': {v: np.zeros((0, 4), np.int32)}, '3-1': {}})
    stats = E.parse_info_stats({'1-3': {1: 'cuts_clust:2\ncuts_extra:3\nno_extra_cuts:1\nt_total:  0.5s,  1.500%\n'}, '3-1': {}})
    assert stats['1-3'] == {'cuts_clust': [2], 'cuts_extra': [3], 'no_extra_cuts': [1], 't_total': [1.5]}
    text = E.format_report([('smartvidcrop', {'1-3': (1.0, 2.0, 1.5)}, stats, 0)])
    assert text.splitlines()[1].split(',')[1:12] == ['1.000', '2.000', '1.500', '1.500', '1.500', '-1.000', '-1.000',
                                                     '2.000', '2.000', '3.000', '3.000']


def test_entry_points_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        return
    import pytest
    from retargetvid_amd import _lib
    with pytest.raises(_lib.SvcError):
        S.smart_vid_crop({'fr': 30.0, 'frame_count': 8, 'w': 64, 'h': 36,
                          'frames': np.zeros((8, 36, 64, 3), np.uint8), 'trans_inds': [0, 8]},
                         S.sc_init_crop_params(), save_vid=False)
    with pytest.raises(NotImplementedError):
        S.smart_vid_crop('movie.mp4', S.sc_init_crop_params(), save_vid=False, engine=object())


def test_decode_hand_off_adapter(monkeypatch):
    """ingest.read_video_cv2 against a stand-in for OpenCV: BGR -> RGB, frame count, shot list -> trans_inds."""
    import sys, types
    from retargetvid_amd import ingest
    rng = np.random.RandomState(0)
    bgr = rng.randint(0, 256, (7, 6, 8, 3)).astype(np.uint8)

    class Cap:
        def __init__(self, path): self.i = 0
        def isOpened(self): return True
        def get(self, prop): return {5: 25.0, 7: 7}[prop]
        def read(self):
            if self.i >= len(bgr): return False, None
            self.i += 1
            return True, bgr[self.i - 1]
        def release(self): pass

    fake = types.SimpleNamespace(VideoCapture=Cap, CAP_PROP_FPS=5, CAP_PROP_FRAME_COUNT=7)
    monkeypatch.setitem(sys.modules, 'cv2', fake)
    v = ingest.read_video_cv2('x.mp4', shot_detector=lambda fr: [3, 3, 0, 99])
    assert v['fr'] == 25.0 and v['frame_count'] == 7 and (v['w'], v['h']) == (8, 6)
    assert np.array_equal(v['frames'], bgr[..., ::-1]) and v['trans_inds'] == [0, 3, 7]
    assert ingest.video_dict(bgr, 30)['trans_inds'] == [0, 7]
    S.set_video_reader(lambda path, CP: v)
    try:
        with pytest.raises(Exception) as ei:          # reaches the device path, which needs a GPU here
            S.smart_vid_crop('x.mp4', S.sc_init_crop_params(), save_vid=False)
        assert not isinstance(ei.value, NotImplementedError)
    finally:
        S.set_video_reader(None)


def test_round_plan_of_the_cluster_filter_with_held_maps():
    """svc_cluster_center's planner (host code of the HIP library, no GPU needed): a map blended from its predecessor runs
    one round later; a held map (SVC_MAP_HELD) is skipped; a map blended from a held one runs in round 0 after the blend."""
    import ctypes
    from retargetvid_amd import _lib
    lib = _lib.load()
    B, H = 1, 2

    def plan(flags):
        fl = np.asarray(flags, np.uint8)
        r, b = np.zeros(len(fl), np.int32), np.zeros(len(fl), np.int32)
        vp = ctypes.c_void_p
        n = lib.svc_debug_round_plan(fl.ctypes.data_as(vp), len(fl), r.ctypes.data_as(vp), b.ctypes.data_as(vp))
        return n, r.tolist(), b.tolist()

    assert plan([0, 0, 0]) == (1, [0, 0, 0], [0, 0, 0])
    assert plan([B, B, 0, 0, B, 0]) == (3, [0, 1, 2, 0, 0, 1], [0] * 6)                 # the reference's shot start: 0 -> 1 -> 2
    assert plan([B, 0, H, 0, B, H]) == (2, [0, 1, -1, 0, 0, -1], [0] * 6)               # the chains' last maps left for later
    assert plan([H | B, 0, H | B, 0]) == (1, [-1, 0, -1, 0], [0, 1, 0, 1])              # ... and finished by the next call
    assert plan([H | B, B, 0]) == (2, [-1, 0, 1], [0, 1, 0])                            # a carried chain goes on
    assert plan([H, H]) == (0, [-1, -1], [0, 0])
    assert plan([H | B, H, 0]) == (1, [-1, -1, 0], [0, 0, 0])                           # a held map does not pass a blend on
    # bench.py's one-round schedule: head [A0, A1, B0, B1, C] + own maps 0 (processed), 1, 2 (held), 3 ...
    n, r, b = plan([H | B, 0, H | B, 0, H, 0, H, H, 0, 0])
    assert n == 1 and r == [-1, 0, -1, 0, -1, 0, -1, -1, 0, 0] and b == [0, 1, 0, 1, 0, 0, 0, 0, 0, 0]
    assert plan([]) [0] == 0
    assert lib.svc_debug_round_plan(np.ones(300, np.uint8).ctypes.data_as(ctypes.c_void_p), 300,
                                    np.zeros(300, np.int32).ctypes.data_as(ctypes.c_void_p),
                                    np.zeros(300, np.int32).ctypes.data_as(ctypes.c_void_p)) < 0      # chain too long


def test_video_path_selection_and_scenes_match_the_oracle():
    """retargetvid_amd.smartVidCrop._select_frames_video / the scene fix of detect_shots against the oracle's restatement
    of read_and_segment_video (smartVidCrop.py:379-399, :452-457) on random transition probabilities."""
    from oracle import pipeline_ref as P
    from retargetvid_amd import smartVidCrop as S, transnetv1_handler as T
    rng = np.random.RandomState(2)
    for trial in range(30):
        n = int(rng.randint(3, 400))
        probs = rng.rand(n).astype(np.float32) * (rng.rand(n) < 0.08)
        if trial == 0:
            probs[:] = 0.9                                             # every frame a transition
        if trial == 1:
            probs[:] = 0.0
        rb = int(rng.choice([50, 64, 2000]))
        got = S._select_frames_video(n, n, probs, 0.1, 6, rb)
        ref = P.select_frames_video(n, n, probs, 0.1, 6, rb)
        assert got[0] == ref[0] and got[1] == ref[1] and got[2] == ref[2]
        seg = np.array(T.predictions_to_scenes(probs, threshold=0.1), dtype=np.int32)
        for i in range(len(seg) - 1):
            seg[i][1] = seg[i + 1][0] - 1
        seg[-1][1] = n - 1
        assert np.array_equal(seg, P.scenes_from_probs(probs, 0.1))


def test_transnet_weights_from_tensorflow_variable_names():
    """weights.transnet_from_tf_variables: a dict as exported from the reference's TF1 graph (3rd_party_libs/transnetv1/
    transnetv1_handler.py:25-91: names with ':0', optimiser slots, counters) -> the packer's state dict; wrong
    shapes and missing variables are named."""
    import pytest
    from retargetvid_amd import weights as W
    sd = W.make_transnet_state_dict(3)
    tfv = {k + ':0': v.copy() for k, v in sd.items()}
    tfv['TransNet/dense/kernel/Adam:0'] = np.zeros_like(sd['TransNet/dense/kernel'])
    tfv['TransNet/dense/kernel/Adam_1:0'] = np.zeros_like(sd['TransNet/dense/kernel'])
    tfv['beta1_power:0'] = np.float32(0.9)
    tfv['global_step:0'] = np.int64(1234)
    out = W.transnet_from_tf_variables(tfv)
    assert sorted(out) == sorted(sd) and all(np.array_equal(out[k], sd[k]) for k in sd)
    assert np.array_equal(W.pack_transnet_blob(out), W.pack_transnet_blob(sd))
    assert out['TransNet/SDDCNN_2/DDCNN_1/Conv3D_4/kernel'].shape == (3, 3, 3, 64, 32)      # TF layout: [kt, kh, kw, cin, filters]
    bad = dict(tfv)
    del bad['TransNet/SDDCNN_3/DDCNN_2/Conv3D_8/bias:0']
    with pytest.raises(KeyError, match='Conv3D_8/bias'):
        W.transnet_from_tf_variables(bad)
    bad = dict(tfv)
    bad['TransNet/dense_1/kernel:0'] = np.zeros((2, 256), np.float32)                         # transposed by mistake
    with pytest.raises(ValueError, match='dense_1/kernel'):
        W.transnet_from_tf_variables(bad)


def test_plan_video_bookkeeping_for_the_scheduler():
    """smartVidCrop.plan_video (host only): selection as the oracle's, the all-zero-map rows = the last selected frame of
    every read batch (the reference's off-by-one, smartVidCrop.py:408-453), blend flags ending with two zeros (nothing is
    blended across a video boundary when videos are packed into one stream), errors for inconsistent input."""
    import pytest
    n = 437
    video = dict(fr=25.0, frame_count=n, w=480, h=640, frames=np.zeros((n, 1, 1, 3), np.uint8), trans_inds=[0, 100, 290, n])
    CP = dict(S.sc_init_crop_params(), read_batch=150)
    plan = S.plan_video(video, CP)
    ti, m2o, batches = P.select_frames(n, n, [0, 100, 290, n], CP['skip'], CP['read_batch'])
    assert plan['true_inds'] == ti and plan['map2orig'] == m2o and plan['batches'] == batches
    assert (plan['sal_h'], plan['sal_w']) == P.sal_size(480, 640, 250) == (250, 187)
    zero = np.flatnonzero(plan['zero_map']).tolist()
    assert zero == [first + cnt - 1 for first, cnt in batches] and len(zero) == 3
    assert plan['flags'].tolist() == S.blend_flags(plan['n_sel'], plan['seg_sel']).tolist()
    assert plan['flags'][-2:].tolist() == [0, 0] and plan['flags'][:2].tolist() == [1, 1]
    assert plan['seg'].tolist() == [[0, 99], [100, 289], [290, n - 1]] and plan['seg_sel'][-1][1] == plan['n_sel'] - 1
    with pytest.raises(ValueError):
        S.plan_video(dict(video, trans_inds=[7]), CP)
    with pytest.raises(ValueError):
        S.plan_video(dict(video, trans_inds=None), CP)              # no shots and no shot network
    # A video that BEGINS inside a transition: predictions_to_scenes (smartVidCrop.py:211-228) opens the first scene where the
    # probability first drops below the threshold, so the opening frames are in no scene.  The reference has no check for it and
    # fails in the per-shot interpolation (so does the oracle's restatement: found by tools/soak_video_path.py); here it is a
    # ValueError from the host bookkeeping, before any device work -- and the same for a trans_inds list that does not begin with 0.
    from retargetvid_amd import transnetv1_handler as T
    probs = np.zeros(n, np.float32)
    probs[:100] = 0.6
    probs[177] = probs[179] = 0.5
    seg = np.array(T.predictions_to_scenes(probs, threshold=0.1), dtype=np.int32)
    for i in range(len(seg) - 1):
        seg[i][1] = seg[i + 1][0] - 1
    seg[-1][1] = n - 1
    assert seg[0][0] == 100
    shots = dict(trans_probs=probs, segmentation=seg, trans_inds=T.shots_to_trans_inds(seg, n))
    with pytest.raises(ValueError, match='first scene starts at frame 100'):
        S.plan_video(dict(video, trans_inds=None), CP, shots=shots)
    with pytest.raises(ValueError, match='trans_inds must begin with 0'):
        S.plan_video(dict(video, trans_inds=[5, 100, n]), CP)
    from oracle import temporal_ref as TR
    ti, m2o, _ = P.select_frames_video(n, n, probs, 0.1, CP['skip'], CP['read_batch'])
    oseg = P.scenes_from_probs(probs, 0.1)
    osel = np.array([[m2o[v] for v in row] for row in oseg], dtype=np.int32)
    with pytest.raises((IndexError, ValueError)):                     # the reference's arithmetic on that input
        dxi, dyi = TR.interpolate_centres(list(np.linspace(20, 100, len(ti))), list(np.linspace(30, 60, len(ti))), oseg, osel, ti)
        TR.smoothing(dxi, dyi, oseg, 25.0, dict(P.init_crop_params(), read_batch=150))


def test_lazy_result_dict_reports_lazy_keys_without_building_them():
    """VD of the multi-video job: 'smaps' / 'bbs' are built on demand; len(), iteration, keys(), `in` and bool() must not
    trigger the device-to-host copy of every map (advisor, round 4)."""
    from retargetvid_amd import smartVidCrop as S

    class FakeMaps:
        built = 0

        def permute(self, *a):
            FakeMaps.built += 1
            return self

        def cpu(self):
            return self

        def numpy(self):
            return np.zeros((2, 3, 1), np.uint8)

    vd = S._LazySmaps({'fc': 7, 'smaps_dev': FakeMaps(), 'bbs_np': np.array([[0, 0, 9, 9]])})
    assert len(vd) == 5 and bool(vd) and 'smaps' in vd and 'bbs' in vd
    assert sorted(vd) == sorted(['fc', 'smaps_dev', 'bbs_np', 'smaps', 'bbs']) and sorted(vd.keys()) == sorted(vd)
    assert FakeMaps.built == 0 and not dict.__contains__(vd, 'bbs')
    assert vd['bbs'] == [[0, 0, 9, 9]] and FakeMaps.built == 0
    assert vd['smaps'].shape == (2, 3, 1) and FakeMaps.built == 1
    assert len(vd) == 5 and dict(vd)['smaps'].shape == (2, 3, 1)


def test_lane_rows_follow_the_job():
    from retargetvid_amd import scheduler
    CP = S_init()
    vids = [{'frame_count': 600, 'frames': None}] * 10
    assert scheduler.lane_rows_for(vids[:3], CP, 12) == 128                 # 3 videos of ~103 rows over 12 lanes: the floor
    assert scheduler.lane_rows_for(vids * 40, CP, 4) == 4096                # capped
    assert 128 < scheduler.lane_rows_for(vids * 4, CP, 4) < 4096
    assert scheduler.lane_rows_for([lambda: None], CP, 4) == 4096           # sizes unknown


def S_init():
    from retargetvid_amd import smartVidCrop as S
    return S.sc_init_crop_params()


def test_pillow_reader_decodes_frame_folders_and_multi_frame_files(tmp_path):
    """The decode door (the reference decodes with cv2, smartVidCrop.py:299-335 -- absent here): the reader that
    does run in this image.  A folder of frame images and multi-frame files come back as the ingest_pickle dict with the very
    bytes that were written (lossless formats), RGB, frame rate from the argument or the file, shots from the detector or left
    to the entry point (shots=None: no trans_inds key)."""
    from PIL import Image
    from retargetvid_amd import ingest
    fr = np.random.RandomState(3).randint(0, 256, (9, 36, 64, 3)).astype(np.uint8)
    d = tmp_path / 'clip'
    d.mkdir()
    for i, f in enumerate(fr):
        Image.fromarray(f).save(str(d / ('frame_%04d.png' % i)))
    (d / 'notes.txt').write_text('not a frame')
    v = ingest.read_frames_pillow(str(d), fr=30.0)
    assert np.array_equal(v['frames'], fr) and v['frames'].flags['C_CONTIGUOUS']
    assert (v['fr'], v['frame_count'], v['w'], v['h'], v['trans_inds']) == (30.0, 9, 64, 36, [0, 9])
    assert ingest.read_frames_pillow(str(d), max_frames=4)['frames'].shape[0] == 4
    ims = [Image.fromarray(f) for f in fr]
    ims[0].save(str(tmp_path / 'a.png'), save_all=True, append_images=ims[1:], duration=40)      # APNG, 40 ms per frame
    v = ingest.read_frames_pillow(str(tmp_path / 'a.png'), shots=None)
    assert np.array_equal(v['frames'], fr) and v['fr'] == 25.0 and 'trans_inds' not in v
    ims[0].save(str(tmp_path / 'a.tiff'), save_all=True, append_images=ims[1:])
    v = ingest.read_frames_pillow(str(tmp_path / 'a.tiff'), fr=24.0, shot_detector=lambda f: [4, 4, 0, 99])
    assert np.array_equal(v['frames'], fr) and v['trans_inds'] == [0, 4, 9] and v['fr'] == 24.0
    Image.fromarray(fr[0][:20]).save(str(d / 'frame_9999.png'))                              # a frame of another size
    with pytest.raises(IOError):
        ingest.read_frames_pillow(str(d))
    (tmp_path / 'empty').mkdir()
    with pytest.raises(IOError):
        ingest.read_frames_pillow(str(tmp_path / 'empty'))                                   # a folder without frame images


def test_look_ahead_bounds_the_planners_and_cannot_dead_lock():
    """scheduler.LookAhead (round-5 advisor: the planner threads ran unbounded ahead of the lanes): three producers walk the items
    round-robin like JobScheduler._plan_ahead, one consumer takes them in order with random pauses like the feeder; no item is ever
    produced more than `limit` ahead of the consumer, every item arrives (no dead-lock, also with limit = 1), stop() and fail()
    release producers that wait."""
    import random
    import threading
    import time
    from retargetvid_amd.scheduler import LookAhead
    for limit, n, producers in ((1, 30, 3), (2, 40, 3), (5, 40, 2), (50, 20, 3)):
        look = LookAhead(limit)
        ready = [threading.Event() for _ in range(n)]
        ahead, lock = [], threading.Lock()

        def produce(k):
            for i in range(k, n, producers):
                if not look.admit(i, poll=0.05):
                    return
                with lock:
                    ahead.append(i - look.taken + 1)
                time.sleep(random.random() * 1e-3)
                ready[i].set()
        ths = [threading.Thread(target=produce, args=(k,), daemon=True) for k in range(producers)]
        for t in ths:
            t.start()
        rng = random.Random(limit)
        for i in range(n):                                   # the feeder: ask for item i, wait for it
            look.take()
            assert ready[i].wait(10.0), 'item %d never arrived (limit %d)' % (i, limit)
            time.sleep(rng.random() * 2e-3)
        for t in ths:
            t.join(10.0)
            assert not t.is_alive()
        assert len(ahead) == n and max(ahead) <= limit and look.high_water <= limit
        if limit < n:
            assert look.high_water == min(limit, n)            # the producers DID run ahead as far as they were allowed
    # producers that wait are released by stop() and by fail()
    for release in ('stop', 'fail'):
        look = LookAhead(1)
        res = []
        t = threading.Thread(target=lambda: res.append(look.admit(5, poll=0.05)), daemon=True)
        t.start()
        time.sleep(0.1)
        assert t.is_alive()                                  # item 5 is not admitted while nothing has been taken
        look.stop() if release == 'stop' else look.fail(RuntimeError('x'))
        t.join(5.0)
        assert not t.is_alive() and res == [False]
