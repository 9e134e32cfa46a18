"""-m gpu: BASELINE.json configs 3, 4 and 5 through the product path (the C ABI), each against the oracle on a subset
and through size-independent properties on the whole input.

  config 3  "200-video RetargetVid set, 1:3 and 3:1 targets, frames sharded": the job code (dist.crop_job: shard by
            frame count -> crop_videos -> gather -> result files -> evaluator) on the 8 shortest videos with their REAL
            frame counts (from the annotation fixtures), both ratios, against the oracle pipeline's windows;
            sharding is covered at world 2 on CPUs (tests/test_dist_gloo.py) and is deterministic by construction
  config 4  "1080p frames, batch=128": 128 frames of 1920x1080 resident in HBM through the whole chain in one call
            (four chunks of 32 inside the library); oracle on a subset, batch-independence and filter properties on all
  config 5  "4K frame stream": 2160x3840 frames, down-scale bit-exact against the OpenCV restatement, chain equal to the
            chain on pre-scaled frames, host-fed ingest (pinned, double-buffered) equal to device-resident ingest"""
import os

import numpy as np
import pytest
import torch

from oracle import cv_ref, pipeline_ref as P, tail_ref as T, unisal_ref as U
from retargetvid_amd import dist as D, evaluate as E, smartVidCrop as S, synth

pytestmark = pytest.mark.gpu


def _device_frames(n, h, w, seed, chunk=16):
    v = synth.LazyBlobVideo(n, h, w, seed=seed)
    return torch.cat([v.select(range(i, min(n, i + chunk))) for i in range(0, n, chunk)])


def test_config3_job_real_frame_counts_two_ratios(engine, synthetic_sd, golden_dir, tmp_path):
    torch.set_num_threads(16)
    annots = E.load_annotations(os.path.join(golden_dir, 'retargetvid'))
    by_len = sorted((len(annots[0]['1-3'][v]), v) for v in E.VID_INDS)[:8]
    vids = [v for _, v in by_len]
    counts = [c for c, _ in by_len]
    assert counts[0] == 230 and counts[-1] == 280
    CP = S.sc_init_crop_params()
    videos = {}

    def make(i):
        def build():
            n = counts[i]
            rng = np.random.RandomState(vids[i])
            cuts = sorted(set([0] + [int(c) for c in rng.randint(20, n - 20, rng.randint(1, 4))]))
            videos[i] = dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.LazyBlobVideo(n, seed=vids[i]),
                             trans_inds=cuts + [n])
            return videos[i]
        return build

    allb, st = D.crop_job(make, counts, ['%03d' % v for v in vids], CP, ('1:3', '3:1'), out_dir=str(tmp_path), workers=3,
                          run_name='cfg3')
    assert st['world'] == 1 and st['videos_rank'] == 8 and st['video_frames_rank'] == sum(counts)
    n_diff = n_tot = 0
    for i in range(8):
        host = dict(videos[i], frames=videos[i]['frames'].select(range(counts[i])).cpu().numpy())     # the same pixels, on the host
        ref = P.smart_vid_crop(host, dict(P.init_crop_params(), out_ratio='1:3'), synthetic_sd)
        wf, hf, _ = T.calc_dest_size(640, 360, '3:1')
        exp = {'1:3': np.array(ref['bbs']),
               '3:1': np.array(T.compute_bb(ref['dxs'], ref['dys'], counts[i], 640, 360, 250, 140, wf, hf)[0])}
        for r in ('1:3', '3:1'):
            d = np.abs(allb[r][i] - exp[r])
            assert d.shape == (counts[i], 4) and d.max() <= 1, (vids[i], r, int(d.max()))
            n_diff += int((d.max(1) > 0).sum())
            n_tot += counts[i]
    assert n_diff <= 0.02 * n_tot
    # result files in the reference's format, scored by the evaluator counterpart (192 of the 200 videos are absent)
    rows = open(os.path.join(str(tmp_path), 'cfg3', '%03d_3-1.txt' % vids[0])).read().splitlines()
    assert len(rows) == counts[0] and rows[0].split(',')[0] == '0' and rows[0].split(',')[2] == '640'
    res, _ = E.evaluate(str(tmp_path), os.path.join(golden_dir, 'retargetvid'), out_path=None)
    (run, scores, stats, missing), = res
    assert run == 'cfg3' and missing == 2 * 192 and set(scores) == {'1-3', '3-1'}
    assert 't_total' in stats['1-3'] and len(stats['1-3']['t_total']) == 8


def test_config4_1080p_batch128_whole_chain(engine, synthetic_sd):
    torch.set_num_threads(16)
    CP = P.init_crop_params()
    frames = _device_frames(128, 1080, 1920, seed=77)                       # 796 MB resident in HBM
    assert frames.shape == (128, 1080, 1920, 3) and frames.is_cuda
    flags = np.zeros(128, np.uint8)
    flags[[0, 1, 63, 64]] = 1                                               # two shot starts inside the batch
    small = engine.resize_frames(frames, 140, 250)
    maps = engine.saliency(small)                                           # 4 chunks of 32 inside the library
    engine.threshold_(maps, CP['t_threshold'])
    thr = maps.clone()
    xy = engine.cluster_center_(maps, flags, CP).cpu().numpy()
    # oracle on a subset: the blend chain at the start and three scattered frames
    sub = [0, 1, 2, 40, 64, 65, 127]
    host = frames[sub].cpu().numpy()
    ref_small = np.stack([cv_ref.resize_linear_u8(f, 140, 250) for f in host])
    assert np.array_equal(small[sub].cpu().numpy(), ref_small)
    ref_maps = U.saliency_u8(synthetic_sd, ref_small)
    got_thr = thr[sub].cpu().numpy()
    T.threshold(ref_maps, CP['t_threshold'])
    d = np.abs(np.transpose(ref_maps, (2, 0, 1)).astype(int) - got_thr.astype(int))
    assert ((d > 0) & (d < 120)).mean() < 1e-3 and (d >= 120).mean() < 1e-3       # one grey level, or a threshold flip of one
    # tail on the GPU's own thresholded maps of the subset, against the oracle tail (bit-exact)
    chain = np.ascontiguousarray(np.transpose(got_thr[:3], (1, 2, 0)))
    for i in range(3):
        chain[:, :, i] = T.clustering_filt(chain[:, :, i], CP)
        if i < 2:
            chain[:, :, i + 1] = T.blend_next(chain[:, :, i], chain[:, :, i + 1])
    assert np.array_equal(maps[:3].cpu().numpy(), np.transpose(chain, (2, 0, 1)))
    dx, dy = T.centers(chain, CP)
    for i in range(3):
        assert (dx[i] is None and np.isnan(xy[i, 0])) or (xy[i, 0] == dx[i] and xy[i, 1] == dy[i])
    # properties on all 128: batch independence (maps without a blended predecessor), filter only removes / closes
    for i in (40, 100, 127):
        one = thr[i:i + 1].clone()
        xy1 = engine.cluster_center_(one, None, CP).cpu().numpy()
        assert torch.equal(one[0], maps[i]) and np.array_equal(xy1[0], xy[i], equal_nan=True)
    assert torch.equal(engine.saliency(small[96:128]), engine.saliency(small)[96:128])      # chunk position does not matter
    t = thr.cpu().numpy()
    m = maps.cpu().numpy()
    plain = [i for i in range(128) if i == 0 or not flags[i - 1]]
    closed = np.stack([cv_ref.morph_close_5x5(t[i]) for i in plain])
    assert (m[plain] <= closed).all()
    ok = ~np.isnan(xy[:, 0])
    assert ok.sum() >= 120 and (xy[ok, 0] >= 0).all() and (xy[ok, 0] <= 249).all() and (xy[ok, 1] <= 139).all()


def test_config5_4k_frames_resize_and_chain(engine, synthetic_sd):
    CP = P.init_crop_params()
    frames = _device_frames(6, 2160, 3840, seed=5, chunk=2)                 # 149 MB
    small = engine.resize_frames(frames, 140, 250)
    host = frames[:2].cpu().numpy()
    ref_small = np.stack([cv_ref.resize_linear_u8(f, 140, 250) for f in host])
    assert np.array_equal(small[:2].cpu().numpy(), ref_small)               # 2160x3840 -> 140x250, bit-exact
    noise = torch.randint(0, 256, (2, 2160, 3840, 3), dtype=torch.uint8, device='cuda')      # white noise: every tap matters
    ref_n = np.stack([cv_ref.resize_linear_u8(f, 140, 250) for f in noise.cpu().numpy()])
    assert np.array_equal(engine.resize_frames(noise, 140, 250).cpu().numpy(), ref_n)
    maps = engine.saliency(small)
    ref_maps = U.saliency_u8(synthetic_sd, ref_small)
    d = np.abs(maps[:2].permute(1, 2, 0).cpu().numpy().astype(int) - ref_maps.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    # the whole entry point on a 4K video: device-resident frames and host frames (pinned, double-buffered ingest) agree
    video = dict(fr=30.0, frame_count=6, w=3840, h=2160, frames=frames, trans_inds=[0, 6])
    CPs = dict(S.sc_init_crop_params(), out_ratio='1:1', skip=1)
    VD, _ = S.smart_vid_crop(video, CPs, save_vid=False, engine=engine)
    VH, _ = S.smart_vid_crop(dict(video, frames=frames.cpu().numpy()), CPs, save_vid=False, engine=engine)
    VP, _ = S.smart_vid_crop(dict(video, frames=frames.cpu().pin_memory()), CPs, save_vid=False, engine=engine)   # pinned source: no staging copy
    assert VD['bbs'] == VH['bbs'] == VP['bbs'] and len(VD['bbs']) == 6 and (VD['h_process'], VD['w_process']) == (140, 250)
    b = np.array(VD['bbs'])
    assert (b[:, 2] - b[:, 0] == 2160).all() and (b[:, 3] - b[:, 1] == 2160).all() and (b[:, 0] >= 0).all() and (b[:, 2] <= 3840).all()
