"""Points per map (N after the threshold) of bench.py's synthetic batch, and the tail kernels' time on them (GPU box helper)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, weights
from oracle import pipeline_ref as P
eng = ops.Engine(weights.make_synthetic_state_dict(0))
CP = P.init_crop_params()
sg = os.environ.get('SIGMA')
fr = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=int(os.environ.get('SEED', 100)), **dict((dict(sigma=tuple(float(x) for x in sg.split(','))) if sg else {}), **(dict(n_blobs=int(os.environ['N_BLOBS'])) if os.environ.get('N_BLOBS') else {})))).cuda()
small = eng.resize_frames(fr, 140, 250)
maps = eng.saliency(small)
eng.threshold_(maps, int(os.environ.get('THRESH', CP['t_threshold'])))
n = (maps != 0).flatten(1).sum(1).cpu().numpy()
print('N per map: min %d mean %.0f max %d' % (n.min(), n.mean(), n.max()), n[:8])
flags = np.zeros(32, np.uint8); flags[:2] = 1
for tag in ('flags', 'none'):
    fl = flags if tag == 'flags' else np.zeros(32, np.uint8)
    for _ in range(3):
        m = maps.clone(); eng.cluster_center_(m, fl, CP)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        m = maps.clone(); eng.cluster_center_(m, fl, CP)
    torch.cuda.synchronize()
    print('%s tail %s: %.3f ms' % (os.environ.get('TAG', ''), tag, (time.perf_counter() - t) / 10 * 1e3))
m = maps.clone(); eng.cluster_center_(m, np.zeros(32, np.uint8), CP); torch.cuda.synchronize()
for i in range(0, 32, 6):
    st = eng.cluster_state(i, 35000)
    h = st['hdr']
    print('  map %2d: N=%5d clusters %3d  stamps (us, 100 MHz wall clock): k_sort %.1f | k_tree built %.1f hierarchy %.1f chosen %.1f | '
          'k_finish %.1f | k_prim %.1f' % (i, st['n'], h[4], h[8] / 100.0, h[13] / 100.0, h[9] / 100.0, h[10] / 100.0, h[11] / 100.0, h[12] / 100.0))
    print('           k_core stamps (us): phase 1 (thread per point, inner ring) %.1f, phase 2 (wavefront per point) %.1f, end %.1f' % (h[5] / 100.0, h[6] / 100.0, h[7] / 100.0))
    if h[23]:
        print('           k_tree_par stamps (us): loaded %.1f nearest-greater %.1f clusters %.1f jumped %.1f tops %.1f | rows %.1f stabilities %.1f labels+kept %.1f' % tuple(
            x / 100.0 for x in list(h[25:30]) + [h[13], h[9], h[10]]))
    if h[16]:
        print('           k_prim_lvl: %d rounds, %d rises; setup %.1f us, phases (us): rise %.1f extract %.1f probe %.1f commit %.1f mark %.1f' % (
            h[16], h[17], h[24] / 100.0, h[18] / 100.0, h[19] / 100.0, h[20] / 100.0, h[21] / 100.0, h[22] / 100.0))

