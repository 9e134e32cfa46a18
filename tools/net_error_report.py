"""GPU box helper: error of every network tap (HIP vs oracle) for the non-carrier and the reference-initialised
checkpoints at the three geometries -- the numbers behind the tolerances of tests/test_gpu_parity.py."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unisal_ref as U
from retargetvid_amd import ops, weights
torch.set_num_threads(8)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'unisal_golden2.npz'))
stats = {k[3:]: g[k] for k in g.files if k.startswith('bn/')}
cks = {'nc': weights.make_synthetic_state_dict(3, carrier=False), 'ri': weights.make_reference_init_state_dict(7, stats),
       'carrier': weights.make_synthetic_state_dict(0)}
for ck, sd in cks.items():
    eng = ops.Engine(sd)
    for gname in ('16x9', '4x3', 'port'):
        frames = g['frames_' + gname]
        h, w = frames.shape[1:3]
        NH, NW = U.get_optimal_out_size((h, w))
        maps = eng.saliency(torch.from_numpy(frames).cuda()).cpu().numpy()
        taps = {}
        ref = U.saliency_u8(sd, frames, taps)
        for i in range(frames.shape[0]):
            t = taps['frames'][i]
            row = []
            for which, key, shape in ((ops.TAP_FEAT4X, 'feat_4x', (NH // 8, NW // 8, 64)), (ops.TAP_FEAT2X, 'feat_2x', (NH // 16, NW // 16, 160)),
                                      (ops.TAP_FEAT1X, 'feat_1x', (NH // 32, NW // 32, 1296)), (ops.TAP_POSTCNN, 'post_cnn', (NH // 32, NW // 32, 256)),
                                      (ops.TAP_DEC, 'dec', (NH // 8, NW // 8, 64)), (ops.TAP_PRE, 'pre', (h, w))):
                got = eng.tap(which, i, shape)
                r = t[key][0].permute(1, 2, 0).numpy() if key != 'pre' else t[key][0].numpy()
                if key == 'feat_1x':
                    got = got[:, :, :1280]
                d = np.abs(got - r)
                rel = d / (np.abs(r) + 1e-3 * np.abs(r).max())
                row.append('%s max %.1e mean %.1e p99.9 %.1e rms(r)/max %.2f' % (key, d.max() / np.abs(r).max(), d.mean() / np.abs(r).max(), np.percentile(d, 99.9) / np.abs(r).max(), np.sqrt((r.astype(np.float64) ** 2).mean()) / np.abs(r).max()))
            du = np.abs(maps[i].astype(int) - ref[:, :, i].astype(int))
            print(ck, gname, i, 'u8 diff px %.4f%% max %d | ' % (100 * (du > 0).mean(), du.max()) + ' | '.join(row), flush=True)
    eng.close()
