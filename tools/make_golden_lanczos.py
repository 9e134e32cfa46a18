"""tests/golden/lanczos_golden.npz: Pillow (container version) LANCZOS outputs on seeded inputs.
Run from the repo root:  python tools/make_golden_lanczos.py"""
import numpy as np
import PIL
import PIL.Image

rng = np.random.RandomState(7)
out = {'pillow_version': np.array(PIL.__version__)}
for i, (h, w, oh, ow) in enumerate([(140, 250, 256, 416), (70, 125, 256, 416), (187, 250, 288, 384), (90, 60, 45, 100)]):
    a = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if i == 1:
        a[:, :, :] = (np.linspace(0, 255, w)[None, :, None]).astype(np.uint8)      # smooth ramp
    out['in_%d' % i] = a
    out['out_%d' % i] = np.asarray(PIL.Image.fromarray(a).resize((ow, oh), PIL.Image.LANCZOS))
np.savez_compressed('tests/golden/lanczos_golden.npz', **out)
print('ok')
