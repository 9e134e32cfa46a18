#!/usr/bin/env python3
"""Soak run of the reference's VIDEO path (a video dict without trans_inds: TransNet V1 inside the ingest decides shots and
selection, smartVidCrop.py:234-556) against the oracle pipeline fed the ORACLE's transition probabilities: random lengths (20 ... 330
frames), frame shapes, hard cuts at random places, read batches from 30 frames to "the whole video" (the fr - 5 frame overlap between
batches), skip, frame rates 24 ... 60, several synthetic TransNet checkpoints (their cuts are arbitrary; what is checked is that both
sides derive the same selection, scenes and windows from them).  Pass: |dP| <= 1e-4 on every frame; where no probability of the video
lies within 1e-3 of the threshold: identical selection, scenes and scene rows, windows within +-1 px of the oracle's (a difference of
the windows at all is printed).  SOAK_SHOT_NET=diff replaces TransNet on BOTH sides by a frame-difference detector (diff_probs below) that
finds the videos' hard cuts: the multi-scene bookkeeping of the video path (after-cut selection, scenes, scene rows, blend flags) against
the oracle's.  python tools/soak_video_path.py [videos] [seed]   (GPU box; ~13 s per video with the oracle's TransNet, ~3 s with diff)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cv_ref, pipeline_ref as P, transnet_ref as R
from retargetvid_amd import ops, smartVidCrop as S, synth, transnetv1_handler as Hd, weights
TOL = 1e-4


def diff_probs(arr):
    """A stand-in shot detector for the soak (`kind='diff'`): the "transition probability" of a frame is a function of the mean absolute
    difference of 3 x 4 block means to the frame before it -- so the hard cuts of the synthetic videos ARE found (the random TransNet weights report one
    scene per video) and the multi-scene bookkeeping of the video path is exercised; a frame behind an all-zero frame (the zero head of
    the first read batch's array) reads 0.  The same function serves both sides: the product calls it through DiffShotNet below."""
    a = np.asarray(arr, np.float64)
    n = len(a)
    blocks = a.reshape(n, 3, 9, 4, 12, 3).mean(axis=(2, 4))          # 3 x 4 block means per channel of the 27 x 48 frame
    d = np.zeros(n, np.float64)
    if n > 1:
        d[1:] = np.abs(blocks[1:] - blocks[:-1]).mean(axis=(1, 2, 3))
        d[1:][a[:-1].reshape(n - 1, -1).max(axis=1) == 0] = 0.0
    return np.clip((d - 1.0) / 6.0, 0.0, 1.0).astype(np.float32)      # (a blob that wraps round the frame's edge counts as a cut too: 3 - 7 scenes per video)


class DiffShotNet:
    """The interface smartVidCrop.detect_shots / transnetv1_handler.video_transition_probs use of a shot network (eng, predict_video,
    predict_frames, close), answering with diff_probs: the device still down-scales the frames to 48 x 27 and the read-batch / overlap
    arrays are built by the product's code."""
    _own = True

    def __init__(self, eng):
        self.eng = eng

    def predict_video(self, frames, keep=None):
        return diff_probs(frames.cpu().numpy() if torch.is_tensor(frames) else frames)

    predict_frames = predict_video

    def close(self):
        pass


def soak(n_videos, seed, say=lambda m: print(m, flush=True), kind='transnet'):
    rng = np.random.RandomState(seed)
    usd = weights.make_synthetic_state_dict(0)
    eng = ops.Engine(usd)
    nets = {}
    bad = near = n_frames = n_cuts = n_rejected = 0
    worst_p = 0.0
    worst_w = 0
    t0 = time.time()
    try:
        for k in range(n_videos):
            tsd_seed = int(rng.randint(0, 3))
            if kind == 'diff':
                tsd_seed = -1
                if tsd_seed not in nets:
                    nets[tsd_seed] = (DiffShotNet(eng), None)
            elif tsd_seed not in nets:
                tsd = weights.make_transnet_state_dict(tsd_seed)
                nets[tsd_seed] = (Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=tsd), tsd)
            net, tsd = nets[tsd_seed]
            predict = diff_probs if kind == 'diff' else (lambda a, tsd=tsd: R.predict_video(tsd, a))
            n = int(rng.choice([rng.randint(20, 60), rng.randint(60, 160), rng.randint(160, 330)]))
            h, w = [(90, 160), (90, 160), (120, 160), (160, 90), (360, 640)][rng.randint(0, 5)]
            frames = synth.blob_frames(n, h, w, seed=int(rng.randint(0, 10**6)))
            cuts = sorted(set(int(c) for c in rng.randint(2, max(3, n - 2), rng.randint(0, 4))))
            for j, c in enumerate(cuts):                           # hard cuts: flipped / mirrored / other content behind them
                how = (j + k) % 3
                if how == 0:
                    frames[c:] = frames[c:][:, ::-1]
                elif how == 1:                                      # (not the negative: a bright background is salient everywhere, and the
                    frames[c:] = frames[c:][:, :, ::-1]              # oracle's O(N^2) clustering of a 35 000-point map takes minutes)
                else:
                    frames[c:] = synth.blob_frames(n - c, h, w, seed=int(rng.randint(0, 10**6)))
            fr = float(rng.choice([24.0, 25.0, 30.0, 29.97, 50.0, 60.0]))
            over = dict(read_batch=int(rng.choice([400, 100, 64, 30, 150])), skip=int(rng.choice([6, 6, 3, 9])),
                        out_ratio=str(rng.choice(['1:3', '3:1', '1:1', '9:16'])), hdbscan_min=5)
            video = dict(fr=fr, frame_count=n, w=w, h=h, frames=frames)
            tag = 'video %3d: %3d frames %dx%d fr %g cuts %s transnet %d %s' % (k, n, w, h, fr, cuts, tsd_seed, over)
            small = np.stack([cv_ref.resize_linear_u8(f, 27, 48) for f in frames])
            probs = Hd.video_transition_probs(None, small, fr, over['read_batch'], predict=predict)
            try:
                VD, res = S.smart_vid_crop(video, dict(S.sc_init_crop_params(), **over), save_vid=False, engine=eng, shot_net=net)
            except ValueError as e:
                # a video that begins inside a transition (synthetic TransNet weights report one wherever they like): its opening frames
                # are in no scene; the reference's arithmetic fails on it too (tests/test_host_logic.py) -- the oracle must agree
                first = int(P.scenes_from_probs(probs, S.TRANS_THRESHOLD)[0][0])
                agree = 'first scene starts at frame' in str(e) and first != 0
                say('%s: %s (oracle: first scene at frame %d): %s' % (tag, e, first, 'both reject it' if agree else 'MISMATCH'))
                bad += 0 if agree else 1
                n_rejected += 1
                continue
            dp = float(np.abs(np.asarray(VD['trans_probs']) - probs).max())
            worst_p = max(worst_p, dp)
            n_frames += n
            ok = dp <= TOL
            if np.abs(probs - S.TRANS_THRESHOLD).min() <= 1e-3:     # a probability on the threshold: the two sides may cut differently
                near += 1
                say('%s: a probability within 1e-3 of the threshold (|dP| %.2g): selection not compared' % (tag, dp))
                bad += 0 if ok else 1
                continue
            ref = P.smart_vid_crop(video, dict(P.init_crop_params(), **over), usd, trans_probs=probs, trans_threshold=S.TRANS_THRESHOLD)
            same_sel = list(VD['true_inds']) == list(ref['true_inds']) and list(VD['inds_to_orig']) == list(ref['inds_to_orig']) and \
                np.array_equal(VD['segmentation'], ref['segmentation']) and np.array_equal(VD['segmentation_sel'], ref['segmentation_sel'])
            d = int(np.abs(np.asarray(VD['bbs'], np.int64) - np.asarray(ref['bbs'], np.int64)).max())
            worst_w = max(worst_w, d)
            n_cuts += len(ref['segmentation']) - 1
            ok = ok and same_sel and d <= 1
            if not ok or d > 0:
                say('%s: |dP| %.2g, selection / scenes %s, window difference %d px: %s' % (tag, dp, 'equal' if same_sel else 'DIFFER', d, 'ok' if ok else 'MISMATCH'))
            elif k % 10 == 0:
                say('%s: %d scenes, |dP| %.2g, identical' % (tag, len(ref['segmentation']), dp))
            bad += 0 if ok else 1
    finally:
        for net, _ in nets.values():
            net.close()
        eng.close()
    say('%d videos, %d frames, %d cuts found by the oracle: %d mismatching videos, %d with a probability on the threshold, %d rejected by both sides (they begin '
        'inside a transition), largest |dP| %.2g, largest window difference %d px, %.0f s' % (n_videos, n_frames, n_cuts, bad, near, n_rejected, worst_p, worst_w, time.time() - t0))
    return dict(videos=n_videos, frames=n_frames, mismatches=bad, near_threshold=near, largest_dp=worst_p, largest_window_difference_px=worst_w)


if __name__ == '__main__':
    torch.set_num_threads(int(os.environ.get('SOAK_THREADS', 16)))
    r = soak(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 5, kind=os.environ.get('SOAK_SHOT_NET', 'transnet'))
    sys.exit(1 if r['mismatches'] else 0)
