"""-m gpu: the drop-in entry points end to end against the oracle pipeline, the operator
boundary, result files and the evaluator on the reference's published results."""
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_ref as P, unisal_ref as U
from retargetvid_amd import evaluate as E, smartVidCrop as S, synth, unisal_handler

pytestmark = pytest.mark.gpu


def _video(n, seed, trans):
    return dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.blob_frames(n, 360, 640, seed=seed),
                trans_inds=trans)


def test_smart_vid_crop_matches_oracle_boxes(engine, synthetic_sd, tmp_path):
    torch.set_num_threads(8)
    for ratio, seed, trans in (('1:3', 3, [0, 40, 90]), ('3:1', 4, [0, 90])):
        video = _video(90, seed, trans)
        CP = S.sc_init_crop_params()
        CP['out_ratio'] = ratio
        VD, res = S.smart_vid_crop(video, CP, save_vid=False, engine=engine)
        ref = P.smart_vid_crop(video, dict(P.init_crop_params(), out_ratio=ratio), synthetic_sd)
        assert VD['fc'] == 90 and VD['true_inds'] == ref['true_inds']
        assert np.array_equal(VD['segmentation_sel'], ref['segmentation_sel'])
        got, exp = np.array(VD['bbs']), np.array(ref['bbs'])
        assert got.shape == exp.shape == (90, 4)
        assert np.abs(got - exp).max() <= 1                       # north_star: crop windows to +-1 px
        # u8 saliency maps: <= 1 level on < 0.1 % of the pixels (the reference's [H,W,n] layout)
        d = np.abs(VD['smaps'].astype(int) - ref['smaps'].astype(int))
        assert VD['smaps'].shape == ref['smaps'].shape and (d > 0).mean() < 5e-3
        assert res['result'] == 'smart cropped' and res['info'].startswith(' (360x640)->(140x250)->')
        for k in ('t__read_sal_det', 't__thresh', 't__clustering', 't__center_of_mass', 't__bb', 't_total'):
            assert '%' in res[k]
        p = S.write_results(str(tmp_path / 'default_config'), '001', ratio, VD, res)
        rows = open(p).read().splitlines()
        assert len(rows) == 90 and all(len(r.split(',')) == 4 for r in rows)
    VD2, _ = S.smartVidCrop(video, CP, save_vid=False, engine=engine)            # deterministic
    assert VD2['bbs'] == VD['bbs']


def test_other_frame_shapes_match_oracle_boxes(engine, synthetic_sd):
    """Portrait and 4:3 sources: other saliency sizes (250x140, 187x250), other network inputs (416x256, 288x384),
    other tile / patch edge cases in every kernel."""
    torch.set_num_threads(8)
    for (h, w, ratio, seed) in ((640, 360, '1:1', 11), (480, 640, '9:16', 12)):
        video = dict(fr=25.0, frame_count=36, w=w, h=h, frames=synth.blob_frames(36, h, w, seed=seed), trans_inds=[0, 14, 36])
        CP = S.sc_init_crop_params()
        CP['out_ratio'] = ratio
        VD, _ = S.smart_vid_crop(video, CP, save_vid=False, engine=engine)
        ref = P.smart_vid_crop(video, dict(P.init_crop_params(), out_ratio=ratio), synthetic_sd)
        assert (VD['h_process'], VD['w_process']) == (ref['h_process'], ref['w_process'])
        got, exp = np.array(VD['bbs']), np.array(ref['bbs'])
        assert got.shape == exp.shape == (36, 4) and np.abs(got - exp).max() <= 1
        d = np.abs(VD['smaps'].astype(int) - ref['smaps'].astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 5e-3


def test_pickle_door_and_error_conventions(engine, tmp_path):
    import pickle
    video = _video(30, 6, [0, 30])
    p = str(tmp_path / 'v.pkl')
    with open(p, 'wb') as fp:
        pickle.dump(video, fp)
    CP = S.sc_init_crop_params()
    CP['out_ratio'] = '1:3'
    VD, _ = S.smart_vid_crop(p, CP, save_vid=False, engine=engine)
    assert len(VD['bbs']) == 30
    with pytest.raises(ValueError):
        S.smart_vid_crop(dict(video, trans_inds=[0]), CP, save_vid=False, engine=engine)   # no scenes
    with pytest.raises(NotImplementedError):
        S.smart_vid_crop('clip.mp4', CP, save_vid=False, engine=engine)


def test_best_settings_match_oracle(engine, synthetic_sd):
    """use_best_settings=True (ISM'21 set): threshold 90, HDBSCAN 5/3 on maps shrunk by 4, sum-weighted cluster
    choice, nearest-shrunk centres, focus stability, Savitzky-Golay smoothing."""
    torch.set_num_threads(8)
    video = _video(90, 8, [0, 50, 90])
    CP = S.sc_init_crop_params(use_best_settings=True)
    CP['out_ratio'] = '1:3'
    VD, res = S.smart_vid_crop(video, CP, save_vid=False, engine=engine)
    ref = P.smart_vid_crop(video, dict(P.init_crop_params(True), out_ratio='1:3'), synthetic_sd)
    got, exp = np.array(VD['bbs']), np.array(ref['bbs'])
    assert got.shape == exp.shape == (90, 4) and np.abs(got - exp).max() <= 1
    assert VD['jumps_inds'] == ref['jumps_inds']
    assert 't__focus_stability' in res


def test_operator_boundary_layout(engine, synthetic_sd):
    frames = synth.blob_frames(5, 140, 250, seed=2)
    out = unisal_handler.predictions_from_memory_nuint8_np(engine, frames, [], '')
    assert out.shape == (140, 250, 5) and out.dtype == np.uint8 and out.flags['C_CONTIGUOUS']
    ref = U.saliency_u8(synthetic_sd, frames)
    d = np.abs(out.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    with pytest.raises(IndexError):
        unisal_handler.predictions_from_memory_nuint8_np(engine, frames[:0], [], '')


def test_evaluator_on_gpu_reproduces_published_numbers(golden_dir, tmp_path):
    d = os.path.join(golden_dir, 'retargetvid')
    rows, text = E.evaluate(os.path.join(d, 'results_smartvidcrop.zip'), d, out_path=str(tmp_path / 'eval_current.txt'))
    (run, scores, stats, missing), = rows
    assert run == 'smartvidcrop' and missing == 0
    ref13, ref31 = (48.639, 50.855, 49.935), (70.116, 73.606, 71.428)
    assert all(abs(a - b) < 1e-3 for a, b in zip(scores['1-3'], ref13))      # README.md:57-62
    assert all(abs(a - b) < 1e-3 for a, b in zip(scores['3-1'], ref31))
    assert ',48.639,50.855,49.935,' in open(str(tmp_path / 'eval_current.txt')).read()


def test_multi_ratio_run_equals_separate_calls_and_lazy_frames(engine):
    video = dict(fr=30.0, frame_count=120, w=640, h=360, frames=synth.LazyBlobVideo(120, seed=21),
                 trans_inds=[0, 55, 120])
    CP = S.sc_init_crop_params()
    both = S.smart_vid_crop_ratios(video, CP, ('1:3', '3:1'), engine=engine)
    for ratio in ('1:3', '3:1'):
        VD, _ = S.smart_vid_crop(video, dict(CP, out_ratio=ratio), save_vid=False, engine=engine)
        assert VD['bbs'] == both[ratio][0]['bbs'] and len(VD['bbs']) == 120
    b13, b31 = np.array(both['1:3'][0]['bbs']), np.array(both['3:1'][0]['bbs'])
    assert (b13[:, 2] - b13[:, 0] == 120).all() and (b13[:, 1] == 0).all() and (b13[:, 3] == 360).all()
    assert (b31[:, 3] - b31[:, 1] == 213).all() and (b31[:, 0] == 0).all() and (b31[:, 2] == 640).all()
    # the lazy generator is deterministic and indexable in any order
    a = video['frames'].select([5, 17])
    b = video['frames'].select([17])
    assert torch.equal(a[1], b[0]) and a.shape == (2, 360, 640, 3) and a.dtype == torch.uint8


def test_videos_in_flight_equal_sequential_runs(engine):
    """S.crop_videos returns exactly what sequential calls return: the job-level scheduler (the default: frames of
    consecutive videos packed into full network chunks, retargetvid_amd/scheduler.py) and the round-3 form (worker
    threads, one video + engine + stream each)."""
    CP = S.sc_init_crop_params()
    vids = [_video(40 + 7 * k, 30 + k, [0, 17 + k, 40 + 7 * k]) for k in range(5)]
    seq = [S.smart_vid_crop_ratios(v, CP, ('1:3', '3:1'), engine=engine) for v in vids]
    for packed in (True, False):
        par = S.crop_videos([(lambda v=v: v) for v in vids], CP, ('1:3', '3:1'), workers=3, packed=packed)
        for a, b in zip(seq, par):
            for r in ('1:3', '3:1'):
                assert a[r][0]['bbs'] == b[r][0]['bbs'] and a[r][0]['dx'] == b[r][0]['dx']


def test_videos_in_flight_with_the_tail_inside_the_ingest(engine):
    """S.crop_videos(stream_batch=): several videos in flight, each with its tail inside the ingest (16 maps per call) -- same
    windows as sequential whole-video calls, for both parameter sets (the best-settings set reads the filtered maps back
    for its focus stability)."""
    for best in (False, True):
        CP = S.sc_init_crop_params(use_best_settings=best)
        vids = [_video(60 + 9 * k, 40 + k, [0, 21 + k, 60 + 9 * k]) for k in range(6)]
        seq = [S.smart_vid_crop_ratios(v, CP, ('1:3', '3:1'), engine=engine) for v in vids]
        par = S.crop_videos([(lambda v=v: v) for v in vids], CP, ('1:3', '3:1'), workers=3, stream_batch=16, packed=False)
        for a, b in zip(seq, par):
            for r in ('1:3', '3:1'):
                assert a[r][0]['bbs'] == b[r][0]['bbs'] and a[r][0]['dx'] == b[r][0]['dx']


def test_evaluator_iou_within_1e4_of_the_oracle_on_ten_videos(engine, synthetic_sd, tmp_path):
    """north_star's second tolerance: the evaluator's IoU numbers (retargetvid_eval.py:133-283: per-frame IoU ->
    per-video mean -> per-annotator mean -> worst / best / mean) computed from the GPU path's crop windows and from
    the oracle pipeline's must agree within 1e-4 (IoU as a fraction; 0.01 in the evaluator's percent columns), for both
    target ratios, on ten multi-shot videos scored against fixed synthetic annotations of six annotators.
    +-1 px on a 120-px-wide window moves ONE frame's IoU by 1.6e-2, so this is a far tighter statement about how
    often the windows differ at all than the +-1 px bound; the fraction of differing frames is asserted too."""
    import json
    from oracle import tail_ref as T
    torch.set_num_threads(8)
    n_vid = 10
    vids = E.VID_INDS[:n_vid]
    annots = [{ar: {} for ar in E.ARS} for _ in range(6)]
    got = {ar: {} for ar in E.ARS}
    exp = {ar: {} for ar in E.ARS}
    n_frames = n_diff = 0
    max_d = 0
    for k, v in enumerate(vids):
        n = 54 + 6 * k
        video = dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.blob_frames(n, 360, 640, seed=500 + k),
                     trans_inds=[0, 20 + 2 * k, n] if k % 2 else [0, 15 + k, 37 + k, n])
        both = S.smart_vid_crop_ratios(video, S.sc_init_crop_params(), ('1:3', '3:1'), engine=engine)
        ref = P.smart_vid_crop(video, dict(P.init_crop_params(), out_ratio='1:3'), synthetic_sd)
        ref_bbs = {'1-3': np.array(ref['bbs'])}
        wf, hf, _ = T.calc_dest_size(640, 360, '3:1')        # nothing before the box arithmetic depends on the ratio
        ref_bbs['3-1'] = np.array(T.compute_bb(ref['dxs'], ref['dys'], n, 640, 360, 250, 140, wf, hf)[0])
        rng = np.random.RandomState(800 + k)
        for ar, ratio in (('1-3', '1:3'), ('3-1', '3:1')):
            got[ar][v] = np.array(both[ratio][0]['bbs'], np.int32)
            exp[ar][v] = ref_bbs[ar].astype(np.int32)
            assert got[ar][v].shape == exp[ar][v].shape == (n, 4)
            d = np.abs(got[ar][v] - exp[ar][v]).max(1)
            n_frames += n
            n_diff += int((d > 0).sum())
            max_d = max(max_d, int(d.max()))
            for user in range(6):                               # annotators: the oracle's window, offset, plus a smooth random walk
                if ar == '1-3':
                    x = np.clip(exp[ar][v][:, 0] + rng.randint(-50, 51) + np.cumsum(rng.randn(n) * 2.0), 0, 520).astype(int)
                    annots[user][ar][v] = np.stack([x, np.zeros(n, int), x + 120, np.full(n, 360)], 1).astype(np.int32)
                else:
                    y = np.clip(exp[ar][v][:, 1] + rng.randint(-40, 41) + np.cumsum(rng.randn(n) * 1.5), 0, 147).astype(int)
                    annots[user][ar][v] = np.stack([np.zeros(n, int), y, np.full(n, 640), y + 213], 1).astype(np.int32)
    scores = {}
    for name, boxes in (('gpu', got), ('oracle', exp)):
        gt, mt, index = E.pair_boxes(annots, boxes)
        ious = ops_iou(gt, mt)
        scores[name] = E.aggregate(ious, index)
    report = dict(videos=n_vid, frames=n_frames, frames_with_different_window=n_diff, max_window_difference_px=max_d,
                  scores_percent=scores)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'iou_parity.json'), 'w') as fp:
            json.dump(report, fp, indent=1)
    assert max_d <= 1                                            # north_star: windows within +-1 px
    assert n_diff <= 0.02 * n_frames, report                     # measured: see DESIGN.md section 2
    for ar in E.ARS:
        for a, b in zip(scores['gpu'][ar], scores['oracle'][ar]):
            assert abs(a - b) <= 1e-2, report                    # percent units: 1e-4 as a fraction
        assert 5.0 < scores['gpu'][ar][2] < 95.0                 # the synthetic annotations overlap the windows: a real score


def ops_iou(gt, mt):
    from retargetvid_amd import ops
    return np.asarray(ops.iou_boxes(gt, mt), np.float64)


def test_stream_pipeline_equals_the_oracle_loop(engine):
    """retargetvid_amd.pipeline.StreamPipeline: 9 batches of 32 maps with cuts at random positions (blend chains that
    start anywhere, also across batch boundaries and ring wraps), one tail round per call -- filtered maps and centres
    must be those of the reference's sequential loop (oracle: filter map i, blend it into map i+1 next to a cut)."""
    from oracle import tail_ref as T
    from retargetvid_amd import pipeline as PL
    rng = np.random.RandomState(21)
    n, B = 9 * 32 - 5, 32
    ys, xs = np.mgrid[0:140, 0:250]
    maps = np.zeros((n, 140, 250), np.uint8)
    for i in range(n):
        for _ in range(rng.randint(1, 3)):
            cy, cx, r = rng.randint(10, 130), rng.randint(10, 240), rng.randint(4, 9)
            maps[i][(ys - cy) ** 2 + (xs - cx) ** 2 < r * r] = rng.randint(130, 256)
        maps[i][rng.rand(140, 250) < 0.0005] = 200
    maps[7] = 0                                                              # an empty map inside a chain
    cuts = sorted(set([0, 31, 32, 64, 200]) | set(int(c) for c in rng.choice(n - 3, 9, replace=False)))
    seg = np.array([[cuts[i], (cuts[i + 1] - 1) if i + 1 < len(cuts) else n - 1] for i in range(len(cuts))], np.int32)
    flags = S.blend_flags(n, seg)
    CP = S.sc_init_crop_params()
    # the reference's loop
    ref = np.ascontiguousarray(np.transpose(maps, (1, 2, 0)))
    for i in range(n):
        ref[:, :, i] = T.clustering_filt(ref[:, :, i], CP)
        if i + 1 < n and flags[i]:
            ref[:, :, i + 1] = T.blend_next(ref[:, :, i], ref[:, :, i + 1])
    dx, dy = T.centers(ref, CP)
    # the stream
    out = torch.zeros((n, 140, 250), dtype=torch.uint8, device='cuda')
    pipe = PL.StreamPipeline(engine, CP, 140, 250, batch=B, ring_batches=4, max_span_batches=3, maps_out=out)
    dm = torch.from_numpy(maps).cuda()
    got = {}
    for s0 in range(0, n, B):
        if len(pipe.calls) >= pipe.depth:
            got.update({g: (x, y) for g, x, y in pipe.collect()})
        pipe.submit_maps(dm[s0:s0 + B], flags[s0:s0 + B])
    got.update({g: (x, y) for g, x, y in pipe.finish()})
    assert sorted(got) == list(range(n))
    assert np.array_equal(out.cpu().numpy(), np.transpose(ref, (2, 0, 1)))
    for i in range(n):
        if dx[i] is None:
            assert np.isnan(got[i][0]) and np.isnan(got[i][1])
        else:
            assert got[i] == (dx[i], dy[i]), i
    # and the whole sequence in one call (rounds inside the call) gives the same
    one = torch.from_numpy(maps).cuda()
    xy1 = engine.cluster_center_(one, flags, CP).cpu().numpy()
    assert torch.equal(one, out)
    assert all((np.isnan(xy1[i, 0]) and np.isnan(got[i][0])) or tuple(xy1[i]) == got[i] for i in range(n))


def test_streaming_tail_inside_the_ingest_gives_the_same_windows(engine):
    """smart_vid_crop(stream_batch=) -- threshold, cluster filter, cut blend and centres inside the ingest, 32 / 7 maps at
    a time through pipeline.StreamPipeline -- against the whole-video call: same filtered maps, centres and windows,
    for both parameter sets (the best-settings set clusters on shrunk maps)."""
    video = _video(200, 12, [0, 37, 41, 120, 200])
    for best in (False, True):
        CP = S.sc_init_crop_params(use_best_settings=best)
        CP['out_ratio'] = '1:3'
        VD0, _ = S.smart_vid_crop(video, CP, save_vid=False, engine=engine)
        for sb in (32, 7):
            VD1, _ = S.smart_vid_crop(video, CP, save_vid=False, engine=engine, stream_batch=sb)
            assert torch.equal(VD0['smaps_dev'], VD1['smaps_dev'])
            assert VD0['dx'] == VD1['dx'] and VD0['dy'] == VD1['dy'] and VD0['bbs'] == VD1['bbs']


def test_com_km_false_takes_the_argmax_pixel(engine, synthetic_sd):
    """CP['com_km'] = False (sc_find_center_of_mass with km=False, smartVidCrop.py:1165-1178): the centre is the position
    of the first maximum of the filtered map in raster order -- against the oracle pipeline, both parameter sets (the
    best-settings set clusters on shrunk maps but takes the arg-max on the full-size map)."""
    torch.set_num_threads(8)
    video = _video(60, 15, [0, 25, 60])
    for best in (False, True):
        CP = dict(S.sc_init_crop_params(use_best_settings=best), com_km=False, out_ratio='1:3')
        VD, _ = S.smart_vid_crop(video, CP, save_vid=False, engine=engine)
        ref = P.smart_vid_crop(video, dict(P.init_crop_params(best), com_km=False, out_ratio='1:3'), synthetic_sd)
        assert VD['true_inds'] == ref['true_inds']
        assert all(float(a).is_integer() and float(b).is_integer() for a, b in zip(VD['dxnf'], VD['dynf']))
        assert np.abs(np.array(VD['bbs']) - np.array(ref['bbs'])).max() <= 1


def test_feature_cache_in_temp_path(engine, tmp_path):
    """temp_path (smartVidCrop.py:2244-2256, :2276-2280): the ingest's result is pickled per named video and read back on the
    next call -- other target ratio, same windows as a fresh run; the cached call does not touch the frames."""
    video = dict(_video(60, 22, [0, 31, 60]), name='clip22')
    CP = S.sc_init_crop_params()
    CP['out_ratio'] = '1:3'
    fresh13, _ = S.smart_vid_crop(video, CP, save_vid=False, engine=engine)
    a, _ = S.smart_vid_crop(video, CP, save_vid=False, engine=engine, temp_path=str(tmp_path))
    assert os.path.isfile(os.path.join(str(tmp_path), 'clip22.pkl')) and a['bbs'] == fresh13['bbs']
    CP31 = dict(CP, out_ratio='3:1')
    fresh31, _ = S.smart_vid_crop(video, CP31, save_vid=False, engine=engine)
    no_frames = dict(video, frames=None)                       # the cached analysis is enough
    b, _ = S.smart_vid_crop(no_frames, CP31, save_vid=False, engine=engine, temp_path=str(tmp_path))
    assert b['bbs'] == fresh31['bbs'] and b['true_inds'] == fresh31['true_inds']
    assert np.array_equal(b['smaps'], fresh31['smaps'])
    with pytest.raises(TypeError):                             # another frame selection: the cached analysis is not re-used
        S.smart_vid_crop(no_frames, dict(CP31, skip=4), save_vid=False, engine=engine, temp_path=str(tmp_path))
    assert 'bbs' in b and b.get('bbs') == fresh31['bbs'] and set(['bbs', 'smaps']) <= set(dict(b))     # the lazy keys look present


def test_frame_numbers_reach_the_device_without_a_synchronising_copy(engine):
    """smartVidCrop.device_index: an arithmetic progression is generated on the device, any other list goes through the
    pinned ring (more lists than the ring has slots, longer than a slot); CUDA-tensor frames selected with it are the
    frames selected on the host; the lazily generated video gives the same frames with and without the device index."""
    lists = [[0, 6, 12, 18], [5], [], [3, 4, 5, 6, 7], [0, 6, 12, 13, 14, 20], list(range(0, 9000, 2)) + [8999, 9001],
             [7, 3, 11], [1, 2, 4, 8, 16], [10, 20, 30, 31], [0, 1, 3]]
    for idx in lists:
        got = S.device_index(engine, idx)
        assert got.dtype == torch.int64 and got.is_cuda and got.cpu().tolist() == list(idx)
    frames = torch.from_numpy(synth.blob_frames(24, 90, 160, seed=5)).cuda()
    idx = [0, 6, 7, 8, 14, 23]
    small = S._small_frames(engine, frames, idx, 45, 80, engine.device)
    assert torch.equal(small, engine.resize_frames(frames[idx].contiguous(), 45, 80))
    v = synth.LazyBlobVideo(40, 90, 160, seed=2)
    assert torch.equal(v.select(idx), v.select(idx, index=S.device_index(engine, idx)))
    assert torch.equal(v.select([2, 5, 8, 11]), v.select([2, 5, 8, 11], index=S.device_index(engine, [2, 5, 8, 11])))


@pytest.mark.gpu
def test_threshold_regime_diagnostic(engine, tmp_path):
    """Round-4 verdict, "Next round" 6: the user can tell which regime a checkpoint lives in.  svc_threshold_census counts the
    pixels of the UN-thresholded maps at t - 1 / t / t + 1 inside the fused threshold entry; smart_crop_results carries the
    mean per map and level (`pixels_per_grey_level_at_threshold`), the single-video path from the raw maps, the packed job
    from the per-frame rows the network's last kernel counts (svc_saliency_census_u8: the VIDEO's own figure, whatever shares
    its lane); write_results puts it into <vid>_info.txt (a line without '%': the evaluator skips it)."""
    fr = torch.from_numpy(synth.blob_frames(9, 140, 250, seed=5)).cuda()
    raw = engine.saliency(fr).cpu().numpy().astype(int)
    engine.threshold_census(reset=True)
    engine.saliency(fr, threshold=120)
    engine.saliency(fr[:4], threshold=120)
    c = engine.threshold_census()
    both = np.concatenate([raw, raw[:4]])
    assert (c['maps'], c['below'], c['at'], c['above']) == (13, int((both == 119).sum()), int((both == 120).sum()), int((both == 121).sum()))
    assert engine.threshold_census(reset=True)['maps'] == 13 and engine.threshold_census()['maps'] == 0
    engine.saliency(fr)                                            # the plain entry does not count
    assert engine.threshold_census()['maps'] == 0
    video = _video(60, 11, [0, 31, 60])
    CP = S.sc_init_crop_params()
    VD, res = S.smart_vid_crop(video, dict(CP, out_ratio='1:3'), save_vid=False, engine=engine)
    ppl = res['pixels_per_grey_level_at_threshold']
    assert ppl is not None and 0 < ppl < 2000
    # per-frame rows: what the caller's rows receive is the census of each frame's raw map (added: the caller zeroes)
    rows = torch.zeros((9, 4), dtype=torch.int32, device='cuda')
    engine.saliency(fr, threshold=120, census=rows)
    engine.saliency(fr[:4], threshold=120, census=rows[:4])
    want = np.stack([[(m == 119).sum(), (m == 120).sum(), (m == 121).sum(), 0] for m in raw])
    want[:4] *= 2
    assert np.array_equal(rows.cpu().numpy(), want)
    other = _video(45, 12, [0, 45])                                 # a second video sharing the lane: each reports ITS OWN figure
    ppl2 = S.smart_vid_crop(other, dict(CP, out_ratio='1:3'), save_vid=False, engine=engine)[1]['pixels_per_grey_level_at_threshold']
    job = S.crop_videos([video, other], CP, ('1:3',), workers=1)
    assert job[0]['1:3'][1]['pixels_per_grey_level_at_threshold'] == ppl
    assert job[1]['1:3'][1]['pixels_per_grey_level_at_threshold'] == ppl2 != ppl
    fn = S.write_results(str(tmp_path), 'v', '1:3', VD, res)
    info = open(fn.replace('.txt', '_info.txt')).read()
    assert 'pixels_per_grey_level_at_threshold:%s' % ppl in info


@pytest.mark.gpu
def test_a_folder_of_frame_images_through_the_decode_door(engine, tmp_path):
    """§8 f2: smart_vid_crop(<path>) with a reader installed (set_video_reader) -- here the Pillow reader that runs in this image
    (retargetvid_amd/ingest.py; the reference decodes with cv2, smartVidCrop.py:299-335) -- gives the windows of the call on the
    decoded frames; without trans_inds (shots=None) the entry point runs shot detection itself."""
    from PIL import Image
    from retargetvid_amd import ingest, transnetv1_handler as TH, weights
    video = _video(40, 21, [0, 17, 40])
    d = tmp_path / 'clip_001'
    d.mkdir()
    for i, f in enumerate(video['frames']):
        Image.fromarray(f).save(str(d / ('%05d.png' % i)))
    CP = dict(S.sc_init_crop_params(), out_ratio='1:3')
    ref = S.smart_vid_crop(video, CP, save_vid=False, engine=engine)
    try:
        S.set_video_reader(lambda path, cp: ingest.read_frames_pillow(path, fr=video['fr'], shot_detector=lambda fr: [17]))
        VD, res = S.smart_vid_crop(str(d), CP, save_vid=False, engine=engine)
        assert VD['bbs'] == ref[0]['bbs'] and VD['dx'] == ref[0]['dx'] and res['info'] == ref[1]['info']
        net = TH.ShotTransNet(TH.ShotTransNetParams(), weights=weights.make_transnet_state_dict(0), engine=engine)
        S.set_video_reader(lambda path, cp: ingest.read_frames_pillow(path, fr=video['fr'], shots=None))
        a = S.smart_vid_crop(str(d), CP, save_vid=False, engine=engine, shot_net=net)
        b = S.smart_vid_crop({k: v for k, v in video.items() if k != 'trans_inds'}, CP, save_vid=False, engine=engine, shot_net=net)
        assert a[0]['bbs'] == b[0]['bbs'] and np.array_equal(a[0]['segmentation'], b[0]['segmentation'])
    finally:
        S.set_video_reader(None)


def test_random_videos_and_parameters_every_difference_explained(engine, synthetic_sd):
    """tools/soak_e2e.py on a few videos (the soak proper: profiles/r06_job_x_soak_e2e_explained.txt, 140 videos): random frame
    shapes, cuts, frame rates, skip / read_batch / threshold / target ratio, both parameter sets.  The windows must be within
    north_star's +-1 px of the oracle pipeline's, selection and scenes equal; a video whose centres differ AT ALL is accepted only if
    the raw maps of the two implementations differ by at most one grey level with a pixel on either side of the threshold AND the
    oracle's tail fed the GPU's maps reproduces the GPU's centres and windows exactly."""
    from tools import soak_e2e
    lines = []
    r = soak_e2e.soak(10, 21, engine, synthetic_sd, say=lines.append)
    assert r['mismatches'] == 0 and r['beyond_1px'] == 0, '\n'.join(lines)
    assert r['videos'] == 10 and r['frames'] > 0
