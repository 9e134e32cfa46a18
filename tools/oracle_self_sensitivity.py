"""How stable are the crop windows of the ORACLE ITSELF under another fp32 summation order?

The reference-initialised checkpoint ('ri', tools/iou_parity.py) gives diffuse saliency maps: 11-19 k of 35 000 pixels
above the threshold and ~500 pixels on every grey level next to it.  Two correct fp32 implementations of the network
differ by one grey level on a few 0.1 % of the pixels; this script measures what that does downstream by running the
CPU oracle twice -- PyTorch's oneDNN convolutions and its native ones -- and comparing u8 maps and final windows.
CPU only (run in the build container): python tools/oracle_self_sensitivity.py -> profiles/r03_oracle_self_sensitivity.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import iou_parity as IP                                              # noqa: E402
from oracle import pipeline_ref as P                                 # noqa: E402
from retargetvid_amd import synth                                    # noqa: E402


def run(video, CP, sd, mkldnn):
    stage = {}
    with torch.backends.mkldnn.flags(enabled=mkldnn):
        VD = P.smart_vid_crop(video, CP, sd, stage)
    return np.array(VD['bbs']), stage['thresholded']


def main():
    torch.set_num_threads(8)
    kinds = os.environ.get('SENS_CHECKPOINTS', 'ri').split(',')
    n_vid = int(os.environ.get('SENS_VIDEOS', 3))
    out = {}
    for kind in kinds:
        sd = IP.checkpoint(kind)
        for best in (False, True):
            CP = dict(P.init_crop_params(best), out_ratio='1:3')
            frames = diff = maxd = 0
            px = px_diff = 0
            for k in range(n_vid):
                n = 54 + 6 * k
                video = dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.blob_frames(n, 360, 640, seed=500 + k),
                             trans_inds=[0, 20 + 2 * k, n] if k % 2 else [0, 15 + k, 37 + k, n])
                b1, m1 = run(video, CP, sd, True)
                b2, m2 = run(video, CP, sd, False)
                d = np.abs(b1 - b2).max(1)
                frames += n; diff += int((d > 0).sum()); maxd = max(maxd, int(d.max()))
                px += m1.size; px_diff += int((m1 != m2).sum())
            key = '%s/%s' % (kind, 'best' if best else 'default')
            out[key] = dict(videos=n_vid, frames=frames, frames_with_different_window=diff, max_window_difference_px=maxd,
                            thresholded_pixels_that_differ=px_diff, fraction_of_pixels=round(px_diff / px, 6))
            print(key, out[key], flush=True)
    with open(os.path.join(ROOT, 'profiles', 'r03_oracle_self_sensitivity.json'), 'w') as fp:
        json.dump(dict(what='CPU oracle with oneDNN convolutions vs the same oracle with PyTorch native convolutions (two fp32 summation orders)',
                       results=out), fp, indent=1)


if __name__ == '__main__':
    main()
