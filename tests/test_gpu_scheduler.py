"""-m gpu: the job-level scheduler (retargetvid_amd/scheduler.py) -- the selected frames of many videos as one stream per
lane, full network chunks packed across video boundaries -- must give, for EVERY video, exactly the windows, centres and
filtered maps of smart_vid_crop_ratios on that video alone (reference loop: smartVidCrop.py:2722-2790, :2359-2414)."""
import numpy as np
import pytest
import torch

from retargetvid_amd import scheduler, smartVidCrop as S, synth

pytestmark = pytest.mark.gpu


def _video(n, seed, trans, h=360, w=640, lazy=False, fr=30.0):
    frames = synth.LazyBlobVideo(n, h, w, seed=seed) if lazy else synth.blob_frames(n, h, w, seed=seed)
    return dict(fr=fr, frame_count=n, w=w, h=h, frames=frames, trans_inds=trans)


def _same(a, b, ratios=('1:3', '3:1'), maps=True):
    for r in ratios:
        va, vb = a[r][0], b[r][0]
        assert va['bbs'] == vb['bbs']
        assert va['dx'] == vb['dx'] and va['dy'] == vb['dy']
        assert va['dxs_smooth'] == vb['dxs_smooth'] and va['dxi'] == vb['dxi']
        assert a[r][1]['info'] == b[r][1]['info'] and a[r][1]['params'] == b[r][1]['params']
        # the regime diagnostic is the VIDEO's own in the packed job too (round-5 advisor: it was the lane's job-wide mean)
        assert a[r][1]['pixels_per_grey_level_at_threshold'] == b[r][1]['pixels_per_grey_level_at_threshold']
    if maps:
        assert torch.equal(a[ratios[0]][0]['smaps_dev'], b[ratios[0]][0]['smaps_dev'])
        assert np.array_equal(a[ratios[0]][0]['smaps'], b[ratios[0]][0]['smaps'])


def test_packed_job_equals_sequential_runs_both_parameter_sets(engine, synthetic_sd):
    """Twelve videos of 25 ... 190 frames (5 ... 33 selected frames: most chunks hold several videos), cuts at the start, in
    the middle, two frames before the end; host arrays and on-device generators; 1 / 2 / 4 lanes; both parameter sets."""
    for best in (False, True):
        CP = S.sc_init_crop_params(use_best_settings=best)
        vids = []
        for k in range(12):
            n = 25 + 15 * k
            trans = [[0, n], [0, n // 2, n], [0, 7, n - 9, n], [0, 3, n // 3, n - 4, n]][k % 4]
            vids.append(_video(n, 700 + k, trans, lazy=(k % 3 == 0)))
        seq = [S.smart_vid_crop_ratios(v, CP, ('1:3', '3:1'), engine=engine) for v in vids]
        for lanes in (1, 2, 4):
            js = scheduler.JobScheduler(CP, ('1:3', '3:1'), lanes=lanes, state_dict=synthetic_sd)
            try:
                par = js.run([(lambda v=v: v) for v in vids])
            finally:
                js.close()
            assert js.stats['network_frames'] == sum(len(s['1:3'][0]['true_inds']) - 1 for s in seq)
            for a, b in zip(seq, par):
                _same(a, b)
            if lanes == 1:
                assert js.stats['mean_chunk_fill'] > 0.9           # 12 videos, 246 network frames: 7 full chunks + 1


def test_packed_job_mixed_geometries_long_video_and_small_storage(engine, synthetic_sd):
    """A lane whose storage is smaller than the job (it is drained and replaced between videos), a video longer than the
    storage, videos of another geometry (4:3 and portrait: other map sizes) in the middle of the queue, a video with two
    read batches (two all-zero maps: the reference's off-by-one per read batch)."""
    CP = dict(S.sc_init_crop_params(), read_batch=150)
    vids = [_video(120, 800, [0, 50, 120]), _video(400, 801, [0, 130, 260, 400], lazy=True),
            _video(60, 802, [0, 60], h=480, w=640), _video(48, 803, [0, 20, 48], h=640, w=360),
            _video(90, 804, [0, 33, 90]), _video(31, 805, [0, 31]), _video(200, 806, [0, 100, 200], lazy=True)]
    seq = [S.smart_vid_crop_ratios(v, CP, ('1:3', '3:1'), engine=engine) for v in vids]
    assert sum(1 for v in seq[1]['1:3'][0]['smaps_dev'] if not v.any()) >= 3      # 400 frames / read_batch 150: three zero maps
    js = scheduler.JobScheduler(CP, ('1:3', '3:1'), lanes=2, state_dict=synthetic_sd, lane_rows=48)
    try:
        par = js.run([(lambda v=v: v) for v in vids])
    finally:
        js.close()
    for a, b in zip(seq, par):
        _same(a, b)


def test_packed_job_failures_surface_and_empty_job(engine, synthetic_sd):
    CP = S.sc_init_crop_params()
    js = scheduler.JobScheduler(CP, ('1:3',), lanes=2, state_dict=synthetic_sd)
    try:
        assert js.run([]) == []
        bad = dict(fr=30.0, frame_count=40, w=640, h=360, frames=synth.blob_frames(40, 360, 640, seed=1), trans_inds=[5])
        with pytest.raises(ValueError):
            js.run([_video(40, 1, [0, 40]), bad])
        ok = js.run([_video(40, 1, [0, 40])])                   # the scheduler is usable after a failed job
        assert len(ok[0]['1:3'][0]['bbs']) == 40
    finally:
        js.close()
