"""SVC_GRAPH=1 (captured passes) gives the maps of the direct launches (GPU box helper)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SVC_GRAPH'] = '1'
from retargetvid_amd import ops, synth
fr = torch.from_numpy(synth.blob_frames(8, 140, 250, seed=3)).cuda()
e = ops.Engine(seed=0)
os.environ.pop('SVC_GRAPH')
e2 = ops.Engine(seed=0)
st = torch.cuda.Stream()
out = torch.empty((8, 140, 250), dtype=torch.uint8, device='cuda')
res = []
with torch.cuda.stream(st):
    for _ in range(4):
        e.saliency(fr, out=out)
        res.append(out.clone())
torch.cuda.synchronize()
d = e2.saliency(fr)
print('graph passes equal the direct one:', [bool(torch.equal(r, d)) for r in res])
