"""Time svc_saliency_u8 alone for B=32 on a side stream (GPU box helper)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth
eng = ops.Engine(seed=0)
fr = torch.from_numpy(synth.blob_frames(32, 140, 250, seed=0)).cuda()
out = torch.empty((32, 140, 250), dtype=torch.uint8, device='cuda')
st = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(st):
    for _ in range(5): eng.saliency(fr, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): eng.saliency(fr, out=out)
    torch.cuda.synchronize(); print('%s saliency B=32: %.3f ms' % (os.environ.get('TAG', ''), (time.perf_counter() - t) / 20 * 1e3))
