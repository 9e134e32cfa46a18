"""4 streams of network passes + K streams each running the 3-round blend chain of ONE shot start (maps 0 -> 1 -> 2:
single-workgroup kernels only), all free running: does the chain slow the network?  (GPU box helper)"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, weights
from oracle import pipeline_ref as P
CP = P.init_crop_params()
sd = weights.make_synthetic_state_dict(0)
frames = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=100)).cuda()
NN, NC = 4, 4
engs = [ops.Engine(sd) for _ in range(NN + NC)]
sts = [torch.cuda.Stream() for _ in engs]
small = engs[0].resize_frames(frames, 140, 250)
base = engs[0].saliency(small); engs[0].threshold_(base, CP['t_threshold'])
outs = [torch.empty_like(base) for _ in engs]
chain = [base[:3].clone() for _ in engs]
fl = np.array([1, 1, 0], np.uint8)
torch.cuda.synchronize()
def run(k, nc, chains_per_pass):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k):
        for i in range(NN):
            with torch.cuda.stream(sts[i]): engs[i].saliency(small, out=outs[i])
        for j in range(nc):
            i = NN + j
            with torch.cuda.stream(sts[i]):
                for _ in range(chains_per_pass):
                    chain[i].copy_(base[:3]); engs[i].cluster_center_(chain[i], fl, CP)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (k * NN) * 1e3
run(2, NC, 1)
print('4 network streams alone                       : %.3f ms per pass' % run(10, 0, 1))
for nc in (1, 2, 4):
    print('  + %d chain streams (1 chain per 4 passes each) : %.3f ms per pass' % (nc, run(10, nc, 1)))
