"""-m gpu: the end-to-end tolerance of north_star (crop windows within +-1 px of the reference CPU path, evaluator IoU
within 1e-4) on the checkpoint whose maps look like a TRAINED saliency network's, and the decomposition of every
end-to-end difference into "the network's one-grey-level differences" and nothing else.

  tl  weights.make_trained_like_state_dict: the reference model with its last decoder stage fitted to blob targets in the build
      container (tools/make_golden_unisal3.py, tests/golden/unisal_golden3.npz): ~440 points above the threshold, ~7 pixels
      per grey level next to it
  ri  the reference-initialised checkpoint (diffuse maps: ~500 pixels per grey level; DESIGN.md section 2 reports that +-1 px
      cannot be promised there by ANY fp32 implementation) -- used here for the decomposition: the oracle's tail and host
      stages fed the GPU's OWN u8 saliency maps must reproduce the GPU's windows exactly, so whatever differs end to end
      comes from the maps (reference: smartVidCrop.py:2293-2522)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import pipeline_ref as P, tail_ref as T
from retargetvid_amd import ops, smartVidCrop as S, synth, weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize('variant', [1, 2])
def test_trained_like_checkpoint_windows_within_one_px_and_iou_within_1e4(golden_dir, variant):
    from tools import iou_parity
    sd = weights.make_trained_like_state_dict(golden_dir, variant=variant)
    eng = ops.Engine(sd)
    try:
        for best in (False, True):
            r = iou_parity.measure(eng, sd, n_vid=6, best=best)
            assert r['frames'] == 2 * sum(54 + 6 * k for k in range(6))
            assert r['max_window_difference_px'] <= 1, r                  # north_star: windows within +-1 px
            assert r['fraction_different'] <= 0.02, r
            assert r['largest_score_difference_percent'] <= 1e-2, r       # percent units: 1e-4 as a fraction
    finally:
        eng.close()


def _video(n, seed, trans):
    return dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.blob_frames(n, 360, 640, seed=seed), trans_inds=trans)


@pytest.mark.parametrize('ck', ['ri', 'tl', 'tl2'])
def test_oracle_tail_and_host_stages_on_the_gpus_own_maps_reproduce_its_windows(ck, golden_dir):
    """Every stage behind the saliency maps (threshold, cluster filter, blend, CLOSE, centres -- bit-exact on the device;
    empty-centre fill, focus stability, interpolation, low-pass, LOESS / Savitzky-Golay, boxes -- native host code) against
    the oracle, on the maps the GPU itself produced: identical windows, both parameter sets."""
    torch.set_num_threads(8)
    if ck in ('tl', 'tl2'):
        sd = weights.make_trained_like_state_dict(golden_dir, variant=1 if ck == 'tl' else 2)
    else:
        g = np.load(os.path.join(golden_dir, 'unisal_golden2.npz'))
        sd = weights.make_reference_init_state_dict(7, {k[3:]: g[k] for k in g.files if k.startswith('bn/')})
    eng = ops.Engine(sd)
    try:
        cases = [(True, _video(60 + 12 * k, 640 + k, [0, 25 + k, 60 + 12 * k])) for k in range(3)] + \
                [(False, _video(48, 650, [0, 22, 48]))]                     # default set: one short video (its maps hold 11-19 k points)
        for best, video in cases:
            CP = dict(S.sc_init_crop_params(use_best_settings=best), out_ratio='1:3')
            raw = S.ingest_frames(video, CP, eng)                             # the GPU's u8 saliency maps, before the threshold
            VD = dict(raw)
            VD['smaps'] = np.ascontiguousarray(raw['smaps_dev'].permute(1, 2, 0).cpu().numpy())
            del VD['smaps_dev']
            ref = P.crop_from_maps(VD, dict(P.init_crop_params(best), out_ratio='1:3'))
            got, _ = S.smart_vid_crop(video, CP, save_vid=False, engine=eng)
            assert got['true_inds'] == ref['true_inds']
            assert np.array_equal(got['smaps'], ref['smaps']), 'filtered maps'
            assert got['bbs'] == ref['bbs'], (ck, best, int(np.abs(np.array(got['bbs']) - np.array(ref['bbs'])).max()))
    finally:
        eng.close()
