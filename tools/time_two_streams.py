"""Do network passes of two engines on two streams overlap?  (GPU box helper)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth
e1, e2 = ops.Engine(seed=0), ops.Engine(seed=0)
fr = torch.from_numpy(synth.blob_frames(32, 140, 250, seed=0)).cuda()
o1 = torch.empty((32, 140, 250), dtype=torch.uint8, device='cuda'); o2 = torch.empty_like(o1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(two, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(s1): e1.saliency(fr, out=o1)
        with torch.cuda.stream(s2 if two else s1): e2.saliency(fr, out=o2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
run(True, 3); run(False, 3)
print('two passes, one stream : %.3f ms' % run(False))
print('two passes, two streams: %.3f ms' % run(True))
print('two passes, one stream : %.3f ms' % run(False))
print('two passes, two streams: %.3f ms' % run(True))
