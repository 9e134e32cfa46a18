"""Which kernel family makes a frame's map depend on its batch / engine / run?  (GPU box helper, round 5.)
For every SVC_MX_MASK bit (one family on the split-bf16 pipe at a time), SVC_MX=f32 and the default: the maps and taps of the
same frames (a) twice on one engine, (b) on a second engine, (c) as a sub-batch, at three geometries."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth
from oracle import unisal_ref as U

SETS = [('f32', {'SVC_MX': 'f32'}), ('default', {})] + [('mask %d' % m, {'SVC_MX_MASK': str(m)}) for m in (1, 2, 4, 8, 16)] + \
       [('irb bit %d' % b, {'SVC_MX_MASK': '2', 'SVC_IRB_MX': str(1 << b)}) for b in (0, 1, 3, 4)] + \
       [('default minpx400', {'SVC_DWPW_MIN_PX': '400'})]


def engine(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return ops.Engine(seed=0)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


# state dependence: the same 32 frames after ANOTHER batch went through the engine (tests/test_gpu_configs.py: chunk position)
A = torch.from_numpy(synth.blob_frames(32, 140, 250, seed=1)).cuda()
B = torch.from_numpy(synth.blob_frames(32, 140, 250, seed=2)).cuda()
AB = torch.cat([B, A])
for name, env in SETS:
    e = engine(env)
    try:
        m1 = e.saliency(A).clone()
        t1 = [e.tap(ops.TAP_FEAT4X, 31, (32, 52, 64)), e.tap(ops.TAP_FEAT2X, 31, (16, 26, 160)), e.tap(ops.TAP_FEAT1X, 31, (8, 13, 1296)),
              e.tap(ops.TAP_POSTCNN, 31, (8, 13, 256)), e.tap(ops.TAP_DEC, 31, (32, 52, 64))]
        e.saliency(B)
        m2 = e.saliency(A).clone()
        m3 = e.saliency(AB)[32:].clone()
        t3 = [e.tap(ops.TAP_FEAT4X, 31, (32, 52, 64)), e.tap(ops.TAP_FEAT2X, 31, (16, 26, 160)), e.tap(ops.TAP_FEAT1X, 31, (8, 13, 1296)),
              e.tap(ops.TAP_POSTCNN, 31, (8, 13, 256)), e.tap(ops.TAP_DEC, 31, (32, 52, 64))]
        print('state  %-18s A,B,A %-5s (%d px)   [B|A] %-5s (%d px)  taps 4x 2x 1x pc dec %s  max|d| %s' % (
            name, bool(torch.equal(m1, m2)), int((m1 != m2).sum()), bool(torch.equal(m1, m3)), int((m1 != m3).sum()),
            [bool(np.array_equal(x, y)) for x, y in zip(t1, t3)], ['%.1e' % float(np.abs(x - y).max()) for x, y in zip(t1, t3)]))
    finally:
        e.close()

for shape in [(140, 250)]:
    h, w = shape
    NH, NW = U.get_optimal_out_size((h, w))
    fr = torch.from_numpy(np.random.RandomState(h).randint(0, 256, (9, h, w, 3)).astype(np.uint8)).cuda()
    for name, env in SETS:
        a, b = engine(env), engine(env)
        try:
            m1 = a.saliency(fr).clone()
            t1 = [a.tap(ops.TAP_FEAT4X, 4, (NH // 8, NW // 8, 64)), a.tap(ops.TAP_FEAT2X, 4, (NH // 16, NW // 16, 160)),
                  a.tap(ops.TAP_POSTCNN, 4, (NH // 32, NW // 32, 256)), a.tap(ops.TAP_DEC, 4, (NH // 8, NW // 8, 64))]
            m2 = a.saliency(fr).clone()
            m3 = b.saliency(fr).clone()
            t3 = [b.tap(ops.TAP_FEAT4X, 4, (NH // 8, NW // 8, 64)), b.tap(ops.TAP_FEAT2X, 4, (NH // 16, NW // 16, 160)),
                  b.tap(ops.TAP_POSTCNN, 4, (NH // 32, NW // 32, 256)), b.tap(ops.TAP_DEC, 4, (NH // 8, NW // 8, 64))]
            m4 = a.saliency(fr[2:6]).clone()
            t4 = [a.tap(ops.TAP_FEAT4X, 2, (NH // 8, NW // 8, 64)), a.tap(ops.TAP_FEAT2X, 2, (NH // 16, NW // 16, 160)),
                  a.tap(ops.TAP_POSTCNN, 2, (NH // 32, NW // 32, 256)), a.tap(ops.TAP_DEC, 2, (NH // 8, NW // 8, 64))]
            print('%-9s %-18s rerun %-5s  other engine %-5s taps %s  sub-batch %-5s taps(frame 4) %s' % (
                shape, name, bool(torch.equal(m1, m2)), bool(torch.equal(m1, m3)), [bool(np.array_equal(x, y)) for x, y in zip(t1, t3)],
                bool(torch.equal(m1[2:6], m4)), [bool(np.array_equal(x, y)) for x, y in zip(t1, t4)]))
        finally:
            a.close(); b.close()
