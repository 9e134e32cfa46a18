"""BASELINE config 1 (plumbing, CPU): a DHF1K-001-shaped synthetic video through the oracle's
restatement of smart_vid_crop via the pickle door -> 450 well-formed rows for target 1:3."""
import numpy as np
import torch

from oracle import pipeline_ref as P
from retargetvid_amd import synth


def test_config1_plumbing_oracle(synthetic_sd):
    torch.set_num_threads(8)
    n = 450                                     # results/smartvidcrop/001_1-3.txt has 450 rows
    frames = synth.blob_frames(n, 360, 640, seed=11)
    video = dict(fr=30.0, frame_count=n, w=640, h=360, frames=frames, trans_inds=[0, n])
    CP = P.init_crop_params()
    CP['out_ratio'] = '1:3'
    st = {}
    VD = P.smart_vid_crop(video, CP, synthetic_sd, st)
    assert VD['fc'] == n and len(VD['bbs']) == n and VD['fc_sel'] == len(VD['true_inds'])
    assert (VD['w_final'], VD['h_final']) == (120, 360)
    b = np.array(VD['bbs'])
    assert (b[:, 1] == 0).all() and (b[:, 3] == 360).all() and (b[:, 2] - b[:, 0] == 120).all()
    assert b[:, 0].min() >= 0 and b[:, 2].max() <= 640
    # the forced last selected frame never reaches the network (off-by-one, smartVidCrop.py:696-709)
    assert st['thresholded'][:, :, -1].sum() == 0 and st['centres_raw'][0][-1] is None
    assert VD['dx'][-1] is not None
    # smooth trajectory: the crop window follows the focus (jumps only when the focus changes blob)
    assert np.median(np.abs(np.diff(b[:, 0]))) <= 6
