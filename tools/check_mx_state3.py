"""Do recycled (non-zero) device allocations change an engine's results?  Engines are created, used at several geometries and batch
sizes and destroyed in a loop, so that later engines get hipMalloc blocks with earlier engines' data in them; the maps and taps of one
fixed batch must be identical in every engine.  Per kernel family (GPU box helper, round 5)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth

SETS = [('f32', {'SVC_MX': 'f32'}), ('default', {})] + [('mask %d' % m, {'SVC_MX_MASK': str(m)}) for m in (1, 2, 4, 8, 16)] + [('minpx400', {'SVC_DWPW_MIN_PX': '400'})]
F = torch.from_numpy(synth.blob_frames(11, 140, 250, seed=1)).cuda()
others = [torch.from_numpy(np.random.RandomState(s).randint(0, 256, (n, h, w, 3)).astype(np.uint8)).cuda()
          for s, (n, h, w) in enumerate([(40, 140, 250), (7, 250, 140), (33, 187, 250), (3, 97, 131)])]
TAPS = [(ops.TAP_FEAT4X, (32, 52, 64)), (ops.TAP_FEAT2X, (16, 26, 160)), (ops.TAP_FEAT1X, (8, 13, 1296)), (ops.TAP_POSTCNN, (8, 13, 256)), (ops.TAP_DEC, (32, 52, 64))]
for name, env in SETS:
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    ref, bad = None, []
    try:
        for it in range(8):
            e = ops.Engine(seed=0)
            if it % 2:                                   # half of the engines see other work first
                for o in others[it % 4:]:
                    e.saliency(o)
            m = e.saliency(F).clone()
            taps = [e.tap(w, 10, sh) for w, sh in TAPS]
            for o in others[:1 + it % 3]:
                e.saliency(o)
            m2 = e.saliency(F).clone()
            if ref is None:
                ref = (m, taps)
            else:
                d = [float(np.abs(a - b).max()) for a, b in zip(taps, ref[1])]
                if not torch.equal(m, ref[0]) or not torch.equal(m2, ref[0]) or max(d) > 0:
                    bad.append((it, int((m != ref[0]).sum()), int((m2 != ref[0]).sum()), ['%.1e' % v for v in d]))
            e.close()
    finally:
        for k, val in old.items():
            if val is None: os.environ.pop(k, None)
            else: os.environ[k] = val
    print('%-10s %s' % (name, 'identical in 8 engines' if not bad else 'DIFFERS: %s' % bad), flush=True)
