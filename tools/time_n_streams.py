"""Network passes of N engines on N streams: time per pass (GPU box helper).  argv: N list"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth
NS = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]
NF = int(os.environ.get('FRAMES', '32'))
fr = torch.from_numpy(synth.blob_frames(NF, 140, 250, seed=0)).cuda()
engs = [ops.Engine(seed=0) for _ in range(max(NS))]
outs = [torch.empty((NF, 140, 250), dtype=torch.uint8, device='cuda') for _ in engs]
sts = [torch.cuda.Stream() for _ in engs]
for n in NS:
    def run(k):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(k):
            for i in range(n):
                with torch.cuda.stream(sts[i]): engs[i].saliency(fr, out=outs[i])
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / (k * n) * 1e3
    run(3)
    print('%d streams: %.3f ms per pass (GPU_MAX_HW_QUEUES=%s)' % (n, run(15), os.environ.get('GPU_MAX_HW_QUEUES')))
