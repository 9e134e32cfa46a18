"""tests/golden/unisal_golden3.npz: a TRAINED-LIKE checkpoint and the reference model's outputs on it.

The three checkpoints of rounds 1-3 are two synthetic extremes (the luminance carrier: ~45 pixels per grey level next to
the threshold; reference-initialised `ri`: ~500, a diffuse map) and a flat one (`nc`).  north_star's tolerance (windows
within +-1 px) was written for a trained saliency network, whose maps are peaky.  This script makes one in the build
container: the REFERENCE model (3rd_party_libs/unisal/unisal/model.py, imported through tools/ref_import.py) starts from
the `ri` checkpoint (weights.make_reference_init_state_dict(7) + the BatchNorm statistics the reference model calibrated,
tests/golden/unisal_golden2.npz) and its last decoder stage -- skip_4x, post_upsampling_2, adaptation_salicon,
smoothing_salicon: 124 k of the 3.2 M parameters -- is fitted for a few hundred Adam steps on synth.blob_frames with
Gaussian-blob fixation targets (KL divergence between the target distribution and the model's log-softmax map, the first
term of the reference's training loss, train.py:630-660), the static SALICON path, everything else frozen, BatchNorm in
eval mode.  CPU training is not bit-reproducible elsewhere, so the fitted tensors themselves are the fixture:

  tl/<key>                 the fitted tensors (overlay on the `ri` checkpoint: weights.make_trained_like_state_dict)
  frames_<geom>            u8 frames at the three geometries (16:9, 4:3, portrait), 2 each
  logp_/u8_/taps           the reference model's log-softmax map, u8 map (train.py:1270-1274) and taps, as in golden2
  level_hist_16x9          pixels per grey level of its u8 maps over 24 frames (the threshold's neighbourhood is what decides
                           how many points a one-level difference moves)

Run from the repo root:  python tools/make_golden_unisal3.py

`--variant 2` (tests/golden/unisal_golden4.npz, checkpoint `tl2`): a SECOND trained-like checkpoint, so that the parity claims on
peaky maps do not rest on one fit -- another seed, more of the decoder fitted (skip_2x, upsampling_2 as well: ~0.5 M parameters),
narrower targets (a quarter of each blob's width), 240 steps."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import synth, weights                                   # noqa: E402
from tools.make_golden_unisal2 import GEOMS, NET, N, load, prep              # noqa: E402
from tools.ref_import import load_reference_unisal                           # noqa: E402

VARIANT = 2 if '--variant' in sys.argv and sys.argv[sys.argv.index('--variant') + 1] == '2' else 1
TRAINED = ('skip_4x.', 'post_upsampling_2.', 'adaptation_salicon.', 'smoothing_salicon.')
if VARIANT == 2:
    TRAINED = ('skip_2x.', 'upsampling_2.') + TRAINED
STEPS, BATCH = (300, 4) if VARIANT == 1 else (240, 4)
TSIG = 3.0 if VARIANT == 1 else 4.0                                          # target width = blob width / TSIG
SEED0 = 10000 if VARIANT == 1 else 50000
OUT = 'unisal_golden3.npz' if VARIANT == 1 else 'unisal_golden4.npz'
TAG = 'tl' if VARIANT == 1 else 'tl2'


def targets(seeds, h, w):
    """Fixation-density targets: a Gaussian a third as wide as each blob, weighted by its amplitude, normalised to 1."""
    ys, xs = np.arange(h, dtype=np.float64)[:, None], np.arange(w, dtype=np.float64)[None, :]
    out = []
    for sd in seeds:
        x, y, sig, amp = synth.blob_tracks(1, h, w, seed=sd)
        t = np.zeros((h, w))
        for b in range(len(sig)):
            t += amp[b] * np.exp(-((xs - x[0, b]) ** 2 + (ys - y[0, b]) ** 2) / (2 * (sig[b] / TSIG) ** 2))
        out.append(t / t.sum())
    return torch.from_numpy(np.stack(out)).float()


def main():
    torch.manual_seed(0 if VARIANT == 1 else 1)
    torch.set_num_threads(8)
    net, utils = load_reference_unisal()
    g2 = np.load(os.path.join('tests', 'golden', 'unisal_golden2.npz'))
    stats = {k[3:]: g2[k] for k in g2.files if k.startswith('bn/')}
    sd_ri = weights.make_reference_init_state_dict(7, stats)
    load(net, sd_ri)
    net.eval()                                                 # BatchNorm uses the calibrated statistics, dropout off
    params = []
    for k, p in net.named_parameters():
        p.requires_grad_(k.startswith(TRAINED) and not any(s in k for s in ('DHF1K', 'Hollywood', 'UCFSports')))
        if p.requires_grad:
            params.append(p)
    print('fitting %d tensors, %d parameters' % (len(params), sum(p.numel() for p in params)))
    opt = torch.optim.Adam(params, lr=2e-3)
    h, w = 140, 250
    for step in range(STEPS):
        seeds = [SEED0 + step * BATCH + b for b in range(BATCH)]
        fr = np.stack([synth.blob_frames(1, h, w, seed=sd)[0] for sd in seeds])
        x = torch.stack([prep(f, 256, 416) for f in fr])[:, None]
        pred = net(x, target_size=(h, w), source='SALICON', static=True)[:, 0, 0]      # log-softmax maps [B, h, w]
        t = targets(seeds, h, w)
        loss = (t * (torch.log(t.clamp_min(1e-12)) - pred)).sum((1, 2)).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 25 == 0 or step == STEPS - 1:
            print('step %3d  KL %.4f' % (step, float(loss)))
    net.eval()
    out = {}
    sd = net.state_dict()
    for k in sd:
        if k.startswith(TRAINED) and not any(s in k for s in ('DHF1K', 'Hollywood', 'UCFSports', 'num_batches_tracked', 'dhf1k', 'hollywood', 'ucfsports')):
            if not np.array_equal(sd[k].numpy(), sd_ri[k]):
                out['tl/' + k] = sd[k].detach().numpy().copy()
    print('fitted tensors stored:', len(out), sum(v.size for v in out.values()), 'values')
    taps = {}
    hooks = [net.cnn.features[14].register_forward_hook(lambda m, i, o: taps.__setitem__('feat_2x', o)),
             net.post_cnn.register_forward_hook(lambda m, i, o: taps.__setitem__('post_cnn', o)),
             net.adaptation_salicon.register_forward_hook(lambda m, i, o: taps.__setitem__('adapt', o))]
    with torch.no_grad():
        for gname, (gh, gw) in GEOMS.items():
            frames = synth.blob_frames(N, gh, gw, seed=(40 if VARIANT == 1 else 70) + len(gname))
            out['frames_%s' % gname] = frames
            nh, nw = NET[gname]
            for i in range(N):
                pred = net(prep(frames[i], nh, nw)[None, None], target_size=(gh, gw), source='SALICON', static=True)
                smap = torch.squeeze(pred[:, 0, ...].exp()).numpy()
                smap = (smap / np.amax(smap)) * 255.0                     # train.py:1270-1274
                tag = '%s_%s_%d' % (TAG, gname, i)
                out['u8_' + tag] = smap.astype('uint8')
                out['logp_' + tag] = pred[0, 0, 0].numpy()
                if i == 0:
                    for k, v in taps.items():
                        out['%s_%s' % (k, tag)] = v[0].numpy()
            print(gname, 'u8 map: nonzero %.3f, >=120: %.3f, >=90: %.3f' % ((out['u8_' + tag] > 0).mean(), (out['u8_' + tag] >= 120).mean(),
                                                                         (out['u8_' + tag] >= 90).mean()))
        hist = np.zeros(256, np.int64)
        for k in range(24):
            f = synth.blob_frames(1, 140, 250, seed=(900 if VARIANT == 1 else 950) + k)[0]
            pred = net(prep(f, 256, 416)[None, None], target_size=(140, 250), source='SALICON', static=True)
            smap = torch.squeeze(pred[:, 0, ...].exp()).numpy()
            hist += np.bincount(((smap / np.amax(smap)) * 255.0).astype('uint8').ravel(), minlength=256)
    for hk in hooks:
        hk.remove()
    out['level_hist_16x9'] = hist
    print('pixels per grey level per map, levels 110..130: %.1f ; 80..100: %.1f ; >= 120: %.0f per map' % (
        hist[110:131].mean() / 24, hist[80:101].mean() / 24, hist[120:].sum() / 24))
    path = os.path.join('tests', 'golden', OUT)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()
