"""How much do concurrent tails slow the network?  3 streams run network passes only; K other streams run the clustering
tail only (on thresholded maps), continuously.  (GPU box helper)"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, weights
from oracle import pipeline_ref as P
CP = P.init_crop_params()
sd = weights.make_synthetic_state_dict(0)
frames = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=100)).cuda()
NN, NT = 3, 4
engs = [ops.Engine(sd) for _ in range(NN + NT)]
sts = [torch.cuda.Stream() for _ in engs]
small = engs[0].resize_frames(frames, 140, 250)
base = engs[0].saliency(small); engs[0].threshold_(base, CP['t_threshold'])
outs = [torch.empty_like(base) for _ in engs]
torch.cuda.synchronize()
def run(k, ntail, flags, per_net=1):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k):
        for i in range(NN):
            with torch.cuda.stream(sts[i]): engs[i].saliency(small, out=outs[i])
        for j in range(ntail):
            i = NN + j
            with torch.cuda.stream(sts[i]):
                outs[i].copy_(base); engs[i].cluster_center_(outs[i], flags, CP)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (k * NN) * 1e3
f3 = np.zeros(32, np.uint8); f3[:2] = 1
f1 = np.zeros(32, np.uint8)
run(2, NT, f3)
print('3 network streams alone              : %.3f ms per pass' % run(12, 0, f3))
for nt in (1, 2, 4):
    print('  + %d tail streams (3 rounds each)    : %.3f ms per pass' % (nt, run(12, nt, f3)))
    print('  + %d tail streams (1 round each)     : %.3f ms per pass' % (nt, run(12, nt, f1)))
