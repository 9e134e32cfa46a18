"""Whole batches (resize + network + threshold + tail) on N streams with the host never waiting in between: the device's
capacity for the bench's step when every queue is always full.  argv: N list.  (GPU box helper)"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, weights
from oracle import pipeline_ref as P
NS = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]
CP = P.init_crop_params()
sd = weights.make_synthetic_state_dict(0)
frames = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=100)).cuda()
flags = np.zeros(32, np.uint8)
if os.environ.get('BENCH_NO_BLEND', '0') != '1':
    flags[:2] = 1
engs = [ops.Engine(sd) for _ in range(max(NS))]
sts = [torch.cuda.Stream() for _ in engs]
maps = [torch.empty((32, 140, 250), dtype=torch.uint8, device='cuda') for _ in engs]
COPY = os.environ.get('COPY', '0') == '1'
pinned = [torch.empty((32, 2), dtype=torch.float64).pin_memory() for _ in engs]
evs = [torch.cuda.Event() for _ in engs]
def batch(i):
    with torch.cuda.stream(sts[i]):
        small = engs[i].resize_frames(frames, 140, 250)
        m = engs[i].saliency(small, out=maps[i])
        engs[i].threshold_(m, CP['t_threshold'])
        xy = engs[i].cluster_center_(m, flags, CP)
        if COPY:
            pinned[i].copy_(xy, non_blocking=True)
            evs[i].record(sts[i])
for n in NS:
    def run(k):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(k):
            for i in range(n):
                batch(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / (k * n) * 1e3
    run(2)
    print('%d streams, free running: %.3f ms per batch' % (n, run(12)))
