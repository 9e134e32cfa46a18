"""Where do the one-grey-level differences between two correct fp32 implementations of the saliency network come from?

Round 3's verdict: the GPU's u8 maps differ from the oracle's on 0.26-0.40 % of the pixels (reference-initialised checkpoint
`ri`), the oracle differs from ITSELF under another summation order on "0.11 %" -- is the gap reducible?  The two figures were
not the same quantity (the second counted pixels of THRESHOLDED maps, i.e. only the ~40 % of the pixels above the threshold).
This script measures one quantity -- the fraction of ALL pixels of the u8 map that differ, per checkpoint -- between

  A  the oracle as it stands (PyTorch CPU, oneDNN convolutions, BatchNorm applied after every convolution as the
     reference does)
  B  the same with PyTorch's native convolutions (another summation order, nothing else)
  C  the oracle with BatchNorm FOLDED into the convolution weights (w' = w * gamma / sqrt(var + eps), bias' = beta - mean *
     gamma / sqrt(var + eps)): what every inference engine, this one included, does; mathematically the same network
  G  (on the GPU box, --gpu) the HIP network

CPU part: python tools/flip_sources.py          GPU part: python tools/flip_sources.py --gpu
-> profiles/r04_flip_sources[_gpu].json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import iou_parity as IP                                              # noqa: E402
from oracle import unisal_ref as U                                   # noqa: E402
from retargetvid_amd import synth                                    # noqa: E402


def folded(sd, eps=1e-5):
    """The same checkpoint with every BatchNorm folded into the convolution in front of it."""
    out = {k: np.array(v, np.float32) for k, v in sd.items()}
    for k in list(sd):
        if not k.endswith('running_var'):
            continue
        p = k[:-len('.running_var')]                                  # '<block>.<i>' or '<block>.<i>.bn_SALICON'
        base = p[:-len('.bn_SALICON')] if p.endswith('.bn_SALICON') else p
        head, idx = base.rsplit('.', 1)
        conv = '%s.%d' % (head, int(idx) - 1)
        wkey = conv + '.weight' if (conv + '.weight') in sd else None
        assert wkey is not None, (k, conv)
        g, b = np.asarray(sd[p + '.weight'], np.float64), np.asarray(sd[p + '.bias'], np.float64)
        m, v = np.asarray(sd[p + '.running_mean'], np.float64), np.asarray(sd[k], np.float64)
        s = g / np.sqrt(v + eps)
        out[wkey] = (np.asarray(sd[wkey], np.float64) * s[:, None, None, None]).astype(np.float32)
        shift = b - m * s
        if (conv + '.bias') in sd:                                    # a convolution with its own bias (skip reductions)
            shift = shift + np.asarray(sd[conv + '.bias'], np.float64) * s
            out[conv + '.bias'] = np.zeros_like(out[conv + '.bias'])
        out[p + '.weight'] = np.ones_like(out[p + '.weight'])
        out[p + '.bias'] = shift.astype(np.float32)
        out[p + '.running_mean'] = np.zeros_like(out[p + '.running_mean'])
        out[p + '.running_var'] = np.full_like(out[k], 1.0 - eps)     # 1 / sqrt(var + eps) = 1
    return out


def frac(a, b):
    d = np.abs(a.astype(int) - b.astype(int))
    return float((d > 0).mean()), int(d.max())


def main():
    gpu = '--gpu' in sys.argv
    torch.set_num_threads(8)
    n = int(os.environ.get('FLIP_FRAMES', 12))
    frames = np.concatenate([synth.blob_frames(2, 140, 250, seed=520 + k) for k in range(n // 2)])
    out = {}
    for kind in os.environ.get('FLIP_CHECKPOINTS', 'ri,tl,carrier').split(','):
        sd = IP.checkpoint(kind)
        A = U.saliency_u8(sd, frames)
        row = dict(frames=int(frames.shape[0]), pixels_at_or_above_120=float((A >= 120).mean()))
        if gpu:
            from retargetvid_amd import ops
            eng = ops.Engine(sd)
            G = np.transpose(eng.saliency(torch.from_numpy(frames).cuda()).cpu().numpy(), (1, 2, 0))
            eng.close()
            row['G_vs_A'] = frac(G, A)
            row['G_vs_C'] = frac(G, U.saliency_u8(folded(sd), frames))
        else:
            with torch.backends.mkldnn.flags(enabled=False):
                B = U.saliency_u8(sd, frames)
            C = U.saliency_u8(folded(sd), frames)
            row['B_vs_A_native_convolutions'] = frac(B, A)
            row['C_vs_A_batchnorm_folded'] = frac(C, A)
            thr = lambda m: np.where(m < 120, 0, m)
            row['B_vs_A_on_thresholded_maps_round3_quantity'] = frac(thr(B), thr(A))
        out[kind] = row
        print(kind, row, flush=True)
    dst = os.path.join(ROOT, 'gpurun_out' if gpu and os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else 'profiles',
                       os.environ.get('FLIP_OUT', 'r05_flip_sources%s.json' % ('_gpu' if gpu else '')))
    with open(dst, 'w') as fp:
        json.dump(dict(what='fraction of ALL u8-map pixels that differ (and the largest difference in grey levels) between implementations '
                            'A / B / C / G of the same network, see tools/flip_sources.py', results=out), fp, indent=1)
    print('wrote', dst)


if __name__ == '__main__':
    main()
