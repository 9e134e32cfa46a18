"""Generate tests/golden/unisal_golden.npz by running the REFERENCE model code
(/root/reference/3rd_party_libs/unisal/unisal/{model,models/MobileNetV2,utils}.py) in the
build container on seeded synthetic frames with the seeded synthetic checkpoint of
retargetvid_amd.weights.  The reference pre-processing uses torchvision (absent here),
which only forwards to PIL.Image.resize(LANCZOS) + ToTensor + Normalize
(data.py:1281-1294); that is done with Pillow directly.

Run from the repo root:  python tools/make_golden_unisal.py
"""
import os
import sys

import numpy as np
import PIL.Image
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import synth, weights          # noqa: E402
from tools.ref_import import load_reference_unisal   # noqa: E402

SEED_W, SEED_F, N = 0, 1, 4


def main():
    torch.set_num_threads(1)
    net, utils = load_reference_unisal()
    sd = weights.make_synthetic_state_dict(SEED_W)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(('rnn' in k) or any(s in k for s in ('DHF1K', 'Hollywood', 'UCFSports', 'dhf1k',
               'hollywood', 'ucfsports', 'num_batches_tracked')) for k in missing), missing
    frames = synth.blob_frames(N, 140, 250, seed=SEED_F)
    taps = {}
    hooks = [net.cnn.features[18].register_forward_hook(lambda m, i, o: taps.__setitem__('feat_1x', o)),
             net.cnn.features[14].register_forward_hook(lambda m, i, o: taps.__setitem__('feat_2x', o)),
             net.cnn.features[7].register_forward_hook(lambda m, i, o: taps.__setitem__('feat_4x', o)),
             net.post_cnn.register_forward_hook(lambda m, i, o: taps.__setitem__('post_cnn', o)),
             net.adaptation_salicon.register_forward_hook(lambda m, i, o: taps.__setitem__('adapt', o))]
    out = {'frames': frames}
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    u8 = np.zeros((140, 250, N), np.uint8)
    with torch.no_grad():
        for i in range(N):
            img = PIL.Image.fromarray(frames[i]).resize((416, 256), PIL.Image.LANCZOS)
            x = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)
            x = (x - mean) / std
            pred = net(x[None, None], target_size=(140, 250), source='SALICON', static=True)
            smap = pred[:, 0, ...].exp()
            smap = torch.squeeze(smap).numpy()
            smap = (smap / np.amax(smap)) * 255.0         # train.py:1270-1274
            u8[:, :, i] = smap.astype('uint8')
            out['logp_%d' % i] = pred[0, 0, 0].numpy()
            if i == 0:
                out['input_0'] = x.numpy()
                for k, v in taps.items():
                    out[k + '_0'] = v[0].numpy()
    for h in hooks:
        h.remove()
    out['smaps_u8'] = u8
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                        'unisal_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()
