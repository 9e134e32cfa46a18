"""TransNet V1 on the device (csrc/svc_shot.hip through the C ABI) against the CPU restatement of the reference's graph
(oracle/transnet_ref.py; PARITY UNPINNED against TensorFlow, see its header).  Tolerance: fp32 against fp32 in another
summation order over up to 6 912 products per output -- |dP| <= 1e-4 on the transition probability."""
import numpy as np
import pytest
import torch

from oracle import cv_ref, transnet_ref as R
from retargetvid_amd import ops, transnetv1_handler as Hd, weights

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def net():
    sd = weights.make_transnet_state_dict(0)
    n = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=sd)
    yield n, sd
    n.close()


def _frames(n, seed, smooth=True):
    rng = np.random.RandomState(seed)
    fr = rng.randint(0, 256, (n, 27, 48, 3)).astype(np.uint8)
    if smooth:                                     # a "video": slowly varying frames with two hard cuts
        base = rng.randint(0, 256, (3, 27, 48, 3)).astype(np.float32)
        for i in range(n):
            s = 0 if i < n // 3 else (1 if i < 2 * n // 3 else 2)
            fr[i] = np.clip(base[s] + 8 * np.sin(i / 5.0) + rng.randn(27, 48, 3) * 3, 0, 255).astype(np.uint8)
    return fr


def test_predict_raw_matches_oracle(net):
    n, sd = net
    fr = np.stack([_frames(100, 1), _frames(100, 2, smooth=False)])
    got = n.predict_raw(fr)
    ref = R.forward(sd, fr)
    assert got.shape == (2, 100) and got.dtype == np.float32
    assert np.abs(got - ref).max() <= TOL, float(np.abs(got - ref).max())
    assert ref.max() - ref.min() > 0.05                              # the check is not vacuous: probabilities spread out
    # any window length (predict_raw takes [batch, frames, ...]), odd batch
    fr = _frames(3 * 37, 3).reshape(3, 37, 27, 48, 3)
    assert np.abs(n.predict_raw(fr) - R.forward(sd, fr)).max() <= TOL


@pytest.mark.parametrize('nframes', [7, 100, 130, 451])
def test_predict_video_matches_oracle(net, nframes):
    n, sd = net
    fr = _frames(nframes, 10 + nframes)
    got = n.predict_video(fr)
    ref = R.predict_video(sd, fr)
    assert got.shape == (nframes,)
    assert np.abs(got - ref).max() <= TOL
    assert np.array_equal(n.predict_frames(torch.from_numpy(fr).cuda()), got)      # CUDA input, the reference's alias
    assert np.array_equal(Hd.predictions_to_scenes(got, 0.5), R.predictions_to_scenes(ref, 0.5))


def test_preprocess_is_the_opencv_down_scale(net):
    fr = np.random.RandomState(5).randint(0, 256, (4, 360, 640, 3)).astype(np.uint8)
    got = Hd.shot_preprocess_frames(fr, engine=net[0].eng)
    ref = np.stack([cv_ref.resize_linear_u8(f, 27, 48) for f in fr])
    assert got.shape == (4, 27, 48, 3) and np.array_equal(got, ref)
    assert np.array_equal(Hd.shot_preprocess_frame(fr[1], engine=net[0].eng), ref[1])


KNOB_SETS = [
    ({'SVC_SHOT_MX': 'f32'}, 'f32'),                                   # fp32 MFMA, both operands through LDS (rounds 2-4)
    ({'SVC_SHOT_XCD': '0'}, 'bf16x6'),                                 # the default (split-bf16 planes on v_mfma_f32_16x16x32_bf16) without the XCD-aware tile order
    ({'SVC_SHOT_MX': 'bf16x3'}, 'bf16x3'),                             # three plane pairs (16 significant bits per product)
    ({'SVC_SHOT_M16': '4'}, 'bf16x6'),                                 # 4 / 2 position tiles per wavefront (default 3)
    ({'SVC_SHOT_M16': '2', 'SVC_SHOT_MX': 'bf16x3'}, 'bf16x3'),
]


@pytest.mark.parametrize('knobs,pipe', KNOB_SETS, ids=['+'.join('%s=%s' % kv for kv in k.items()) for k, _ in KNOB_SETS])
def test_kernel_forms_agree(net, knobs, pipe):
    """Every form of the convolution cells computes the same network: the fp32-MFMA form (both operands through LDS), the
    split-bf16 form the handle uses by default (bf16x6: planar split activations, the kw taps kept in the accumulators) with its
    tile knobs, and the three-pair form bf16x3, whose 16-bit products stay inside the same tolerance (measured |dP| 1.4e-5 against
    6e-7).  (Round 6 removed the forms that lost every measurement: SVC_SHOT_FORM 0 / 1, SVC_SHOT_M16=0, SVC_SHOT_PT.)"""
    import os
    n, sd = net
    assert n.matrix_pipe() == 'bf16x6'                                # the default follows SVC_MX
    fr = np.stack([_frames(100, 77), _frames(100, 78, smooth=False)])
    ref = n.predict_raw(fr)
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        other = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=sd)      # the knobs are read when the handle is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        assert other.matrix_pipe() == pipe
        got = other.predict_raw(fr)
    finally:
        other.close()
    oracle = R.forward(sd, fr)
    assert np.abs(got - ref).max() <= TOL and np.abs(got - oracle).max() <= TOL
    if pipe != 'bf16x3':
        assert np.abs(got - oracle).max() <= 5e-6                      # the fp32-class pipes sit two orders inside the tolerance


def test_more_windows_than_one_pass_holds(net):
    """svc_transnet_predict cuts a call into passes of as many windows as its workspace holds (15 on the split-bf16 pipe): 17
    windows in one call are the same numbers, bit for bit, as the same windows in two calls."""
    n, sd = net
    fr = np.stack([_frames(100, 200 + k, smooth=(k % 2 == 0)) for k in range(17)])
    whole = n.predict_raw(fr)
    parts = np.concatenate([n.predict_raw(fr[:9]), n.predict_raw(fr[9:])])
    assert whole.shape == (17, 100) and np.array_equal(whole, parts)
    assert np.abs(whole[[0, 16]] - R.forward(sd, fr[[0, 16]])).max() <= TOL


def test_kept_rows_only_every_layer_on_the_frames_they_depend_on(net):
    """svc_transnet_predict_rows: the caller keeps rows a .. b - 1 of every window (predict_video: the middle 50 of 100,
    transnetv1_handler.py:117-121); a cell reaches 8 frames to either side, so the last four cells run on 50 / 66 / 82 / 98 of the
    100 frames.  The kept rows are BIT FOR BIT those of the full pass -- with every tile knob, both split-bf16 pipes, any row range,
    window lengths other than 100, more windows than one pass holds -- and predict_video is unchanged against the oracle."""
    import os
    n, sd = net
    fr = torch.from_numpy(np.stack([_frames(100, 300 + k, smooth=(k % 2 == 0)) for k in range(3)])).cuda()
    for knobs in ({}, {'SVC_SHOT_M16': '2'}, {'SVC_SHOT_M16': '4', 'SVC_SHOT_XCD': '0'}, {'SVC_SHOT_MX': 'bf16x3'}, {'SVC_SHOT_MX': 'f32'}):
        old = {k: os.environ.get(k) for k in knobs}
        os.environ.update(knobs)
        try:
            other = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=sd)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        try:
            full = other.predict_raw_device(fr)
            for a, b in ((25, 75), (0, 100), (0, 1), (99, 100), (3, 12), (40, 97), (10, 11)):
                part = other.predict_raw_device(fr, rows=(a, b))
                assert torch.equal(part[:, a:b], full[:, a:b]), (knobs, a, b)
            short = fr[:, :37].contiguous()                              # another window length: 37 frames
            assert torch.equal(other.predict_raw_device(short, rows=(5, 30))[:, 5:30], other.predict_raw_device(short)[:, 5:30])
        finally:
            other.close()
    many = torch.from_numpy(np.stack([_frames(100, 400 + k) for k in range(17)])).cuda()      # two passes of the workspace
    assert torch.equal(n.predict_raw_device(many, rows=(25, 75))[:, 25:75], n.predict_raw_device(many)[:, 25:75])
    import ctypes
    out = torch.empty((3, 100), dtype=torch.float32, device='cuda')
    for a, b in ((-1, 50), (50, 50), (60, 40), (0, 101)):
        assert n.eng.lib.svc_transnet_predict_rows(n.eng._h, ctypes.c_void_p(fr.data_ptr()), 3, 100, a, b, ctypes.c_void_p(out.data_ptr()), None) < 0


def test_clone_copies_the_knobs_handle_to_handle_and_unknown_pipes_are_rejected(net):
    """ShotTransNet.clone(): same weights, matrix pipe and kernel knobs on an engine of its own, copied with svc_transnet_config_get /
    _set -- the process environment is neither read nor written; svc_create rejects a misspelt SVC_MX / SVC_SHOT_MX (it used to select
    the fp32 pipe silently) and svc_transnet_config_set a configuration that does not exist."""
    import ctypes, os
    n, sd = net
    env = {k: os.environ.get(k) for k in ('SVC_SHOT_MX', 'SVC_SHOT_M16', 'SVC_SHOT_XCD')}
    os.environ.update(SVC_SHOT_MX='bf16x3', SVC_SHOT_M16='4', SVC_SHOT_XCD='0')
    try:
        a = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=sd)
    finally:
        for k, v in env.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    try:
        assert a.config() == [3, 4, 0] and a.matrix_pipe() == 'bf16x3'
        before = dict(os.environ)
        c = a.clone()                                                  # the environment now says "defaults": the clone must not follow it
        try:
            assert dict(os.environ) == before
            assert c.config() == a.config() and c.matrix_pipe() == 'bf16x3' and c.eng is not a.eng
            fr = np.stack([_frames(100, 501)])
            assert np.array_equal(c.predict_raw(fr), a.predict_raw(fr))
        finally:
            c.close()
        d = n.clone()                                                  # "follow SVC_MX" is pinned to what the original resolved it to
        try:
            assert d.config()[0] == 6 and d.config()[1:] == n.config()[1:]
        finally:
            d.close()
        for cfg in ((5, 3, 1), (6, 0, 1), (6, 5, 1)):
            assert a.eng.lib.svc_transnet_config_set(a.eng._h, (ctypes.c_int32 * 3)(*cfg)) < 0 and a.config() == [3, 4, 0]
    finally:
        a.close()
    for var, val in (('SVC_MX', 'bf16'), ('SVC_MX', 'bf16x9'), ('SVC_SHOT_MX', '1'), ('SVC_SHOT_MX', 'fp32')):
        old = os.environ.get(var)
        os.environ[var] = val
        try:
            with pytest.raises(Exception, match=var):
                ops.Engine(seed=0)
        finally:
            os.environ.pop(var, None) if old is None else os.environ.__setitem__(var, old)


def test_the_planners_stay_a_bounded_number_of_videos_ahead_of_the_lanes(net):
    """Back-pressure (round-5 advisor): with shot_net= the planner threads build the on-demand videos and run shot detection AHEAD of
    the lanes; they are three to four times faster than the lanes, so unbounded they would materialise the whole job.  A video is
    built only when the feeder is within plan_ahead videos of it; a shot network on a SHARED engine is cloned for every planner."""
    from retargetvid_amd import smartVidCrop as S, synth, scheduler
    n, sd = net
    usd = weights.make_synthetic_state_dict(0)
    CP = dict(S.sc_init_crop_params(), read_batch=64, hdbscan_min=5)
    base = []
    for k in range(3):
        frames = synth.blob_frames(50 + 10 * k, 90, 160, seed=60 + k)
        frames[30:] = frames[30:][:, ::-1]
        base.append(dict(fr=25.0, frame_count=len(frames), w=160, h=90, frames=frames))
    js = scheduler.JobScheduler(CP, ('1:3',), lanes=1, state_dict=usd, shot_net=n, plan_ahead=2)
    ahead = []

    def make(i):
        def f():
            ahead.append(i - js.next_idx + 1)                           # videos built beyond the feeder's, this one included
            return base[i % 3]
        return f
    try:
        out = js.run([make(i) for i in range(24)])
        assert len(out) == 24 and len(ahead) == 24 and max(ahead) <= 2, ahead
        assert js.stats['plan_ahead'] == 2 and 1 <= js.stats['plan_high_water'] <= 2 and js.stats['planners'] == 3
        for i in range(3, 24):
            assert out[i]['1:3'][0]['bbs'] == out[i % 3]['1:3'][0]['bbs']
        assert js._plan_nets[0] is n                                    # n owns its engine: planner 0 runs it, the others run clones
    finally:
        js.close()
    shared = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=sd, engine=n.eng)      # a network on somebody else's engine
    js = scheduler.JobScheduler(CP, ('1:3',), lanes=2, state_dict=usd, shot_net=shared)
    try:
        ref = js.run(base)
        assert len(js._plan_nets) == 3 and all(p is not shared and p.eng is not n.eng for p in js._plan_nets)
        assert js.plan_ahead == 2 + 2 * 3                               # the default: lanes + 2 per planner
        for a, b in zip(ref, out[:3]):
            assert a['1:3'][0]['bbs'] == b['1:3'][0]['bbs']
    finally:
        js.close()
        shared.close()


def test_errors():
    eng = ops.Engine(seed=0)
    try:
        with pytest.raises(ValueError):
            Hd.ShotTransNet(Hd.ShotTransNetParams())                                  # no weights: no silent random network
        out = torch.empty((1, 4), dtype=torch.float32, device='cuda')
        fr = torch.zeros((1, 4, 27, 48, 3), dtype=torch.uint8, device='cuda')
        import ctypes
        rc = eng.lib.svc_transnet_predict(eng._h, ctypes.c_void_p(fr.data_ptr()), 1, 4, ctypes.c_void_p(out.data_ptr()), None)
        assert rc < 0                                                                  # predict before load
        bad = np.zeros(10, np.float32)
        assert eng.lib.svc_transnet_load(eng._h, bad.ctypes.data_as(ctypes.c_void_p), bad.size) < 0
    finally:
        eng.close()


def test_video_batches_with_overlap_and_detect_shots(net):
    """The reference's call site (smartVidCrop.py:248-372): read batches with an overlap taken from the previous batch's
    array (zeros in front of the first), then predictions_to_scenes; smartVidCrop.detect_shots is that chain on the device."""
    from retargetvid_amd import smartVidCrop as S
    n, sd = net
    rng = np.random.RandomState(4)
    base = rng.randint(0, 256, (4, 72, 128, 3)).astype(np.float32)
    fr = np.stack([np.clip(base[min(i // 60, 3)] + rng.randn(72, 128, 3) * 2, 0, 255) for i in range(230)]).astype(np.uint8)
    CP = S.sc_init_crop_params()
    CP['read_batch'] = 100                                            # three read batches, the last one partial
    got = S.detect_shots(fr, 25.0, CP, net=n)
    small = np.stack([cv_ref.resize_linear_u8(f, 27, 48) for f in fr])
    ref = Hd.video_transition_probs(None, small, 25.0, 100, predict=lambda a: R.predict_video(sd, a))
    assert got['trans_probs'].shape == (230,) and np.abs(got['trans_probs'] - ref).max() <= TOL
    from oracle import pipeline_ref as P
    seg = P.scenes_from_probs(ref, 0.1)                                # predictions_to_scenes + the end-of-segment fix (:452-456)
    assert np.array_equal(got['segmentation'], seg)
    assert all(seg[i][1] == seg[i + 1][0] - 1 for i in range(len(seg) - 1)) and seg[-1][1] == 229
    assert got['trans_inds'] == [int(s[0]) for s in seg] + [230]
    # the overlap matters: predicting the batches on their own (no overlap rows) gives other values at the batch heads
    alone = np.concatenate([R.predict_video(sd, small[i:i + 100]) for i in range(0, 230, 100)])
    assert np.abs(alone - ref).max() > 1e-3


def test_smart_vid_crop_runs_shot_detection_inside_the_ingest(net):
    """The reference's video path end to end (smartVidCrop.py:234-556 -> :2218): a video dict WITHOUT trans_inds and
    shot_net= -- TransNet per read batch with the overlap, after-cut selection from its transition probabilities
    (:394-396), scenes with the end-of-segment fix, then saliency -> crop windows -- against the oracle pipeline fed
    with the oracle's transition probabilities.  (Synthetic TransNet weights: the cuts it reports are arbitrary; what is
    checked is that both sides derive the same selection, segmentation and windows from them.)"""
    from oracle import pipeline_ref as P
    from retargetvid_amd import smartVidCrop as S, synth
    n, sd = net
    torch.set_num_threads(8)
    usd = weights.make_synthetic_state_dict(0)
    eng = ops.Engine(usd)
    try:
        frames = synth.blob_frames(150, 90, 160, seed=9)
        frames[60:] = frames[60:][:, ::-1]                              # a hard cut
        video = dict(fr=25.0, frame_count=150, w=160, h=90, frames=frames)
        CP = S.sc_init_crop_params()
        CP.update(read_batch=64, out_ratio='1:3', hdbscan_min=5)        # three read batches (two overlaps)
        VD, res = S.smart_vid_crop(video, CP, save_vid=False, engine=eng, shot_net=n)
        small = np.stack([cv_ref.resize_linear_u8(f, 27, 48) for f in frames])
        probs = Hd.video_transition_probs(None, small, 25.0, 64, predict=lambda a: R.predict_video(sd, a))
        assert np.abs(VD['trans_probs'] - probs).max() <= TOL
        assert np.abs(probs - 0.1).min() > 10 * TOL                      # no probability sits on the threshold
        ref = P.smart_vid_crop(video, dict(P.init_crop_params(), read_batch=64, out_ratio='1:3', hdbscan_min=5), usd,
                               trans_probs=probs)
        assert VD['true_inds'] == ref['true_inds'] and VD['inds_to_orig'] == ref['inds_to_orig']
        assert np.array_equal(VD['segmentation'], ref['segmentation'])
        assert np.array_equal(VD['segmentation_sel'], ref['segmentation_sel'])
        assert np.abs(np.array(VD['bbs']) - np.array(ref['bbs'])).max() <= 1
        with pytest.raises(ValueError):
            S.smart_vid_crop(video, CP, save_vid=False, engine=eng)      # no trans_inds and no shot network
    finally:
        eng.close()


def test_packed_job_with_shot_detection_inside_equals_sequential_runs(net):
    """crop_videos / JobScheduler with shot_net=: videos WITHOUT trans_inds (the reference's video path) packed into one
    stream -- TransNet runs on the lane's stream when a video is planned -- give the windows of one smart_vid_crop call per video."""
    from retargetvid_amd import smartVidCrop as S, synth
    n, sd = net
    usd = weights.make_synthetic_state_dict(0)
    eng = ops.Engine(usd)
    try:
        vids = []
        for k in range(4):
            frames = synth.blob_frames(70 + 20 * k, 90, 160, seed=30 + k)
            frames[25 + 5 * k:] = frames[25 + 5 * k:][:, ::-1]          # a hard cut
            vids.append(dict(fr=25.0, frame_count=len(frames), w=160, h=90, frames=frames))
        CP = dict(S.sc_init_crop_params(), read_batch=64, hdbscan_min=5)
        seq = [{r: S.smart_vid_crop(v, dict(CP, out_ratio=r), save_vid=False, engine=eng, shot_net=n) for r in ('1:3', '3:1')} for v in vids]
        par = S.crop_videos(vids, CP, ('1:3', '3:1'), workers=2, state_dict=usd, shot_net=n)
        for a, b in zip(seq, par):
            for r in ('1:3', '3:1'):
                assert a[r][0]['true_inds'] == b[r][0]['true_inds'] and np.array_equal(a[r][0]['segmentation'], b[r][0]['segmentation'])
                assert a[r][0]['bbs'] == b[r][0]['bbs'] and a[r][0]['dx'] == b[r][0]['dx']
    finally:
        eng.close()


def test_a_failure_in_the_planner_threads_reaches_the_caller(net):
    """Shot detection runs ahead of the lanes in the scheduler's planner threads (three networks: shot_net and two clones): a video
    one of them cannot plan raises in the caller's thread, and the scheduler runs the next job."""
    from retargetvid_amd import smartVidCrop as S, synth, scheduler
    n, sd = net
    usd = weights.make_synthetic_state_dict(0)
    CP = dict(S.sc_init_crop_params(), read_batch=64, hdbscan_min=5)
    good = []
    for k in range(3):
        frames = synth.blob_frames(60 + 10 * k, 90, 160, seed=50 + k)
        frames[30:] = frames[30:][:, ::-1]
        good.append(dict(fr=25.0, frame_count=len(frames), w=160, h=90, frames=frames))
    bad = dict(good[1], frames=np.zeros((40, 90, 160, 4), np.uint8))          # four channels: the down-scale refuses it
    js = scheduler.JobScheduler(CP, ('1:3',), lanes=2, state_dict=usd, shot_net=n)
    try:
        with pytest.raises(Exception):
            js.run([good[0], bad, good[2]])
        ref = js.run(good)
        again = js.run(good)
        assert len(js._plan_nets) == 3 and js._plan_nets[0] is n
        for a, b in zip(ref, again):
            assert a['1:3'][0]['bbs'] == b['1:3'][0]['bbs']
        one = S.smart_vid_crop(good[1], dict(CP, out_ratio='1:3'), save_vid=False, engine=js.engines[0], shot_net=n)
        assert one[0]['bbs'] == ref[1]['1:3'][0]['bbs']
    finally:
        js.close()


def test_only_the_windows_whose_rows_are_kept_are_computed(net):
    """predict_video(keep=(a, b)): the reference's call site predicts read_batch + overlap rows per video (mostly the zero tail)
    and keeps the video's rows; the windows behind them are skipped.  Kept rows: bit for bit those of the full computation."""
    n, sd = net
    arr = np.zeros((2025, 27, 48, 3), np.uint8)
    arr[25:25 + 613] = _frames(613, 5)
    dev = torch.from_numpy(arr).cuda()
    full = n.predict_video(dev)
    for a, b in ((25, 638), (0, 50), (49, 51), (600, 2025), (700, 700)):
        part = n.predict_video(dev, keep=(a, b))
        assert part.shape == full.shape and np.array_equal(part[a:b], full[a:b])
    probs = Hd.video_transition_probs(n, torch.from_numpy(_frames(613, 5)).cuda(), 30.0, 2000)
    assert np.array_equal(probs, full[25:638])
