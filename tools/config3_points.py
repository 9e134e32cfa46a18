#!/usr/bin/env python3
"""Points per thresholded map over the RetargetVid-shaped synthetic set (what the clustering kernels see in config 3)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from retargetvid_amd import evaluate as E, ops, smartVidCrop as S, synth, weights
torch.cuda.set_device(0)
annots = E.load_annotations(os.path.join(ROOT, 'tests', 'golden', 'retargetvid'))
vids = E.VID_INDS[:int(sys.argv[1]) if len(sys.argv) > 1 else 200]
eng = ops.Engine(weights.make_synthetic_state_dict(0))
CP = S.sc_init_crop_params()
allp = []
for v in vids:
    n = len(annots[0]['1-3'][v])
    idx = list(range(0, n, 6))
    fr = synth.LazyBlobVideo(n, seed=v).select(idx)
    m = eng.saliency(eng.resize_frames(fr, 140, 250))
    eng.threshold_(m, CP['t_threshold'])
    allp.append((m > 0).flatten(1).sum(1).cpu().numpy())
p = np.concatenate(allp)
print(json.dumps(dict(maps=int(p.size), mean=float(p.mean()), pct={q: int(np.percentile(p, q)) for q in (1, 10, 25, 50, 75, 90, 99, 100)},
                      above_4352=int((p > 4352).sum()), above_8192=int((p > 8192).sum()), empty=int((p == 0).sum()))))
