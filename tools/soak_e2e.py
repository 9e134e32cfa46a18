#!/usr/bin/env python3
"""Soak run of the WHOLE path against the oracle pipeline: random videos (7 ... 260 frames, four frame shapes, cuts anywhere incl.
next to each other and next to the ends, frame rates 24 ... 60), random selection parameters (skip, read_batch), thresholds, target
ratios and both published parameter sets -- smart_vid_crop on the GPU must give the oracle's crop windows within north_star's +-1 px and the same selected frames and scenes.
Every video whose centres differ at all is printed WITH ITS CAUSE: the raw u8 maps of both sides are compared, and a difference
counts as explained only if the maps differ by at most one grey level and at least one of those pixels straddles the threshold (the
documented regime of DESIGN.md 2: two correct fp32 evaluations of the network differ by one grey level on 0.006 - 0.03 % of the
pixels); the tail fed the GPU's own maps must then reproduce the GPU's centres exactly (decomposition).
python tools/soak_e2e.py [videos] [seed]   (SOAK_CHECKPOINT=carrier|nc|tl|tl2; GPU box; the oracle's network runs on the host: ~1 - 3 s per video)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pipeline_ref as P, tail_ref as T, unisal_ref as U
from retargetvid_amd import ops, smartVidCrop as S, synth, weights
RATIOS = ('1:3', '3:1', '9:16', '1:1', '4:5', '2:3', '16:9', '4:3')


def soak(n_videos, seed, eng, sd, say=lambda m: print(m, flush=True)):
    """-> dict(videos, frames, flips, mismatches, beyond_1px, largest_window_difference_px)."""
    rng = np.random.RandomState(seed)
    bad = over1 = n_frames = n_flip_videos = 0
    worst = 0

    def explain(video, CPo, ref, VD):
        """Why the centres differ: raw maps of the selected frames by both implementations -> (pixels that differ, largest difference in
        grey levels, pixels on different sides of the threshold, oracle tail on the GPU's maps reproduces the GPU's centres)."""
        stage = {}
        P.ingest(video, CPo, sd, stage)
        small = np.concatenate(stage['sal_frames']) if stage.get('sal_frames') else np.zeros((0, 1, 1, 3), np.uint8)
        if not len(small):
            return 0, 0, 0, False
        cpu = U.saliency_u8(sd, small)                                     # [h, w, n]
        gpu = eng.saliency(torch.from_numpy(small).cuda()).cpu().numpy()   # [n, h, w]
        cpu = np.transpose(cpu, (2, 0, 1))
        d = np.abs(cpu.astype(np.int32) - gpu.astype(np.int32))
        t = int(CPo['t_threshold'])
        flips = int(((cpu >= t) != (gpu >= t)).sum())
        # decomposition: the oracle's tail on the GPU's maps (placed at the rows the network writes) must give the GPU's centres
        VDo = P.ingest(video, CPo, sd)
        net_rows = [r for first, cnt in P.select_frames(len(video['frames']), video['frame_count'], list(video['trans_inds']), CPo['skip'], CPo['read_batch'])[2]
                    for r in range(first, first + cnt - 1)]
        VDo['smaps'][:, :, net_rows] = np.transpose(gpu, (1, 2, 0))
        again = P.crop_from_maps(VDo, CPo)
        same = list(again['dx']) == list(VD['dx']) and list(again['dy']) == list(VD['dy']) and np.array_equal(np.asarray(again['bbs']), np.asarray(VD['bbs']))
        return int((d > 0).sum()), int(d.max()), flips, same

    t0 = time.time()
    for k in range(n_videos):
        best = bool(rng.rand() < 0.35)
        n = int(rng.choice([rng.randint(7, 30), rng.randint(30, 120), rng.randint(120, 260)]))
        h, w = [(360, 640), (360, 640), (480, 640), (640, 360), (320, 320)][rng.randint(0, 5)]
        cuts = sorted(set([0] + [int(c) for c in rng.randint(1, max(2, n - 2), rng.randint(0, 5))]))
        if rng.rand() < 0.3 and n > 12:
            cuts = sorted(set(cuts + [cuts[-1] + 1, cuts[-1] + 2]))              # cuts next to each other
        cuts = [c for c in cuts if n - c >= 2]
        over = dict(skip=int(rng.choice([6, 6, 3, 4, 9, 12])), read_batch=int(rng.choice([2000, 2000, 90, 40, 25])),
                    t_threshold=int(rng.choice([120, 120, 90, 100, 150])), out_ratio=str(rng.choice(RATIOS)))
        if best:
            over.pop('t_threshold')
        video = dict(fr=float(rng.choice([24.0, 25.0, 30.0, 29.97, 50.0, 60.0])), frame_count=n, w=w, h=h,
                     frames=synth.blob_frames(n, h, w, seed=int(rng.randint(0, 10**6))), trans_inds=cuts + [n])
        tag = 'video %3d: %3d frames %dx%d fr %g cuts %s best=%d %s' % (k, n, w, h, video['fr'], cuts[1:], best, over)
        try:
            VD, res = S.smart_vid_crop(video, dict(S.sc_init_crop_params(use_best_settings=best), **over), engine=eng, save_vid=False)
            err_gpu = None
        except Exception as e:                                    # e.g. no centre in any frame: the oracle must fail the same way
            err_gpu = e
        try:
            ref = P.smart_vid_crop(video, dict(P.init_crop_params(best), **over), sd)
            err_ref = None
        except Exception as e:
            err_ref = e
        if err_gpu is not None or err_ref is not None:
            same = err_gpu is not None and err_ref is not None and type(err_gpu) is type(err_ref)
            say('%s: %s  gpu: %r  oracle: %r' % (tag, 'both fail alike' if same else 'MISMATCH (one side failed)', err_gpu, err_ref))
            bad += 0 if same else 1
            continue
        got, exp = np.asarray(VD['bbs'], np.int64), np.asarray(ref['bbs'], np.int64)
        ok = got.shape == exp.shape == (n, 4) and list(VD['true_inds']) == list(ref['true_inds']) and \
            [list(s) for s in VD['segmentation']] == [list(s) for s in ref['segmentation']]
        d = int(np.abs(got - exp).max()) if got.shape == exp.shape else 10**6
        cd = max(abs(a - b) for a, b in zip(list(VD['dx']) + list(VD['dy']), list(ref['dx']) + list(ref['dy']))) if len(VD['dx']) == len(ref['dx']) else 1e9
        n_frames += n
        worst = max(worst, d)
        over1 += 1 if d > 1 else 0
        if ok and (d > 0 or cd > 0):
            npx, lvl, flips, same = explain(video, dict(P.init_crop_params(best), **over), ref, VD)
            explained = lvl <= 1 and flips >= 1 and same
            ok = explained                                        # (a flip can move a window by more than a pixel: counted and printed, see beyond_1px)
            n_flip_videos += 1 if ok else 0
            say('%s: window difference %d px, centre difference %.3g -- raw maps: %d pixels differ by <= %d grey level(s), %d on different sides '
                  'of the threshold; oracle tail on the GPU maps %s the GPU centres and windows: %s' % (
                      tag, d, cd, npx, lvl, flips, 'reproduces' if same else 'DOES NOT reproduce', ('explained' + (' (BEYOND +-1 px)' if d > 1 else '')) if ok else 'MISMATCH'))
        elif not ok:
            say('%s: MISMATCH (selection / scenes / shapes)' % tag)
        elif k % 10 == 0:
            say('%s: identical' % tag)
        bad += 0 if ok else 1
    say('%d videos, %d frames: %d videos with a threshold flip (every difference explained), %d UNEXPLAINED videos, %d beyond +-1 px, '
        'largest window difference %d px, %.0f s' % (n_videos, n_frames, n_flip_videos, bad, over1, worst, time.time() - t0))
    return dict(videos=n_videos, frames=n_frames, flips=n_flip_videos, mismatches=bad, beyond_1px=over1, largest_window_difference_px=worst)


if __name__ == '__main__':
    torch.set_num_threads(int(os.environ.get('SOAK_THREADS', 16)))
    ck = os.environ.get('SOAK_CHECKPOINT', 'carrier')          # carrier (the benchmark's) | nc | tl | tl2 (tools/iou_parity.checkpoint)
    if ck == 'carrier':
        sd_ = weights.make_synthetic_state_dict(0)
    else:
        from tools.iou_parity import checkpoint
        sd_ = checkpoint(ck)
    r = soak(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 11, ops.Engine(sd_), sd_)
    sys.exit(1 if r['mismatches'] else 0)
