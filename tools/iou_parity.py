"""End-to-end window / IoU parity of the GPU path against the oracle pipeline for a checkpoint and a parameter set
(north_star: windows within +-1 px, evaluator IoU within 1e-4).  Used by tests/test_gpu_pipeline.py (carrier
checkpoint) and, as a script on the GPU box, for the checkpoints whose activations look like a real network's:

  python tools/iou_parity.py            -> profiles/r03_iou_parity_ri.json (gpurun: gpurun_out/r03_iou_parity_ri.json)

  nc  weights.make_synthetic_state_dict(3, carrier=False): random weights, no luminance carrier
  ri  weights.make_reference_init_state_dict(7, calibrated BatchNorm statistics): the reference constructor's own init"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def checkpoint(kind):
    from retargetvid_amd import weights
    if kind == 'carrier':
        return weights.make_synthetic_state_dict(0)
    if kind == 'nc':
        return weights.make_synthetic_state_dict(3, carrier=False)
    if kind in ('tl', 'tl2'):                             # trained-like: the reference model fitted to blob targets (golden3 / golden4)
        return weights.make_trained_like_state_dict(os.path.join(ROOT, 'tests', 'golden'), variant=1 if kind == 'tl' else 2)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'unisal_golden2.npz'))
    return weights.make_reference_init_state_dict(7, {k[3:]: g[k] for k in g.files if k.startswith('bn/')})


def measure(engine, sd, n_vid=10, best=False, seed0=500, n0=54):
    """-> report dict: frames, frames whose window differs, largest difference (px), evaluator scores of both paths
    against fixed synthetic annotations of six annotators (retargetvid_eval.py's aggregation)."""
    import torch
    from oracle import pipeline_ref as P, tail_ref as T
    from retargetvid_amd import evaluate as E, ops, smartVidCrop as S, synth
    torch.set_num_threads(8)
    vids = E.VID_INDS[:n_vid]
    annots = [{ar: {} for ar in E.ARS} for _ in range(6)]
    got = {ar: {} for ar in E.ARS}
    exp = {ar: {} for ar in E.ARS}
    n_frames = n_diff = max_d = 0
    centre_diff = 0.0
    per_video = []
    for k, v in enumerate(vids):
        n = n0 + 6 * k
        video = dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.blob_frames(n, 360, 640, seed=seed0 + k),
                     trans_inds=[0, 20 + 2 * k, n] if k % 2 else [0, 15 + k, 37 + k, n])
        both = S.smart_vid_crop_ratios(video, S.sc_init_crop_params(use_best_settings=best), ('1:3', '3:1'), engine=engine)
        ref = P.smart_vid_crop(video, dict(P.init_crop_params(best), out_ratio='1:3'), sd)
        ref_bbs = {'1-3': np.array(ref['bbs'])}
        wf, hf, _ = T.calc_dest_size(640, 360, '3:1')        # nothing before the box arithmetic depends on the ratio
        ref_bbs['3-1'] = np.array(T.compute_bb(ref['dxs'], ref['dys'], n, 640, 360, 250, 140, wf, hf)[0])
        cd = max(abs(a - b) for a, b in zip(list(both['1:3'][0]['dx']) + list(both['1:3'][0]['dy']), list(ref['dx']) + list(ref['dy'])))
        centre_diff = max(centre_diff, float(cd))
        rng = np.random.RandomState(800 + k)
        vd = 0
        for ar, ratio in (('1-3', '1:3'), ('3-1', '3:1')):
            got[ar][v] = np.array(both[ratio][0]['bbs'], np.int32)
            exp[ar][v] = ref_bbs[ar].astype(np.int32)
            assert got[ar][v].shape == exp[ar][v].shape == (n, 4)
            d = np.abs(got[ar][v] - exp[ar][v]).max(1)
            n_frames += n
            n_diff += int((d > 0).sum())
            vd = max(vd, int(d.max()))
            for user in range(6):                               # annotators: the oracle's window, offset, plus a smooth random walk
                if ar == '1-3':
                    x = np.clip(exp[ar][v][:, 0] + rng.randint(-50, 51) + np.cumsum(rng.randn(n) * 2.0), 0, 520).astype(int)
                    annots[user][ar][v] = np.stack([x, np.zeros(n, int), x + 120, np.full(n, 360)], 1).astype(np.int32)
                else:
                    y = np.clip(exp[ar][v][:, 1] + rng.randint(-40, 41) + np.cumsum(rng.randn(n) * 1.5), 0, 147).astype(int)
                    annots[user][ar][v] = np.stack([np.zeros(n, int), y, np.full(n, 640), y + 213], 1).astype(np.int32)
        max_d = max(max_d, vd)
        per_video.append(dict(video=v, frames=n, max_window_difference_px=vd))
    scores = {}
    for name, boxes in (('gpu', got), ('oracle', exp)):
        gt, mt, index = E.pair_boxes(annots, boxes)
        scores[name] = E.aggregate(np.asarray(ops.iou_boxes(gt, mt), np.float64), index)
    dscore = max(abs(a - b) for ar in E.ARS for a, b in zip(scores['gpu'][ar], scores['oracle'][ar]))
    return dict(videos=n_vid, best_settings=bool(best), frames=n_frames, frames_with_different_window=n_diff,
                fraction_different=round(n_diff / max(n_frames, 1), 5), max_window_difference_px=max_d,
                largest_centre_difference_saliency_px=centre_diff, largest_score_difference_percent=dscore,
                scores_percent=scores, per_video=per_video)


def main():
    import torch
    from retargetvid_amd import ops
    out = {}
    n_vid = int(os.environ.get('PARITY_VIDEOS', 10))
    for kind in os.environ.get('PARITY_CHECKPOINTS', 'tl,ri,nc').split(','):
        sd = checkpoint(kind)
        eng = ops.Engine(sd)
        for best in (False, True):
            key = '%s/%s' % (kind, 'best' if best else 'default')
            try:
                out[key] = measure(eng, sd, n_vid, best)
                r = out[key]
                print('%-12s frames %d different %d (%.3f %%) max %d px, centre diff %.3g, score diff %.2e %%' % (
                    key, r['frames'], r['frames_with_different_window'], 100 * r['fraction_different'],
                    r['max_window_difference_px'], r['largest_centre_difference_saliency_px'], r['largest_score_difference_percent']), flush=True)
            except Exception as e:                       # e.g. no centre found in any frame with this checkpoint
                out[key] = dict(error=repr(e))
                print(key, 'ERROR', repr(e), flush=True)
        eng.close()
    dst = os.path.join(ROOT, 'gpurun_out' if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else 'profiles', os.environ.get('PARITY_OUT', 'r05_iou_parity.json'))
    with open(dst, 'w') as fp:
        json.dump(out, fp, indent=1)
    print('wrote', dst)


if __name__ == '__main__':
    main()
