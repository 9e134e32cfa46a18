"""A/B of the channel-group fused block (SVC_CGB=1) against the default un-fused 8x13 blocks: output difference and time of
svc_saliency_u8 at B = 32 (GPU box helper)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth
fr = torch.from_numpy(synth.blob_frames(32, 140, 250, seed=0)).cuda()
outs = {}
for v in ('0', '1'):
    os.environ['SVC_CGB'] = v
    eng = ops.Engine(seed=0)
    out = torch.empty((32, 140, 250), dtype=torch.uint8, device='cuda')
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for _ in range(5): eng.saliency(fr, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): eng.saliency(fr, out=out)
        torch.cuda.synchronize(); print('SVC_CGB=%s saliency B=32: %.3f ms' % (v, (time.perf_counter() - t) / 30 * 1e3), flush=True)
    outs[v] = out.clone()
d = (outs['0'].int() - outs['1'].int()).abs()
print('u8 maps: max |diff| %d, differing pixels %d of %d' % (int(d.max()), int((d > 0).sum()), d.numel()))
