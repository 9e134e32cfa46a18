"""Run-to-run reproducibility of the WHOLE per-batch path with several batches sharing the chip: N engines on N streams, each step =
down-scale -> network with the fused threshold -> cluster filter / centres on the SAME 32 frames; per step the down-scaled frames, the
thresholded maps, the filtered maps and the centres are compared with a single-stream reference, so a difference is attributed to
the first stage that shows it.   python tools/soak_pipeline_concurrent.py [N] [ITERS]    (GPU box; SVC_MX selects the matrix pipe)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, scheduler, smartVidCrop as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
CP = S.sc_init_crop_params()
fr = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=100, n_blobs=2, sigma=(30.0, 44.0))).cuda()
flags = np.zeros(32, np.uint8); flags[:2] = 1
engs = [ops.Engine(seed=0) for _ in range(N)]
sts = scheduler.lane_streams(torch.device('cuda', 0), N)


def step(e):
    small = e.resize_frames(fr, 140, 250)
    thr = e.saliency(small, threshold=CP['t_threshold'])
    filt = thr.clone()
    xy = e.cluster_center_(filt, flags, CP)
    return small, thr, filt, xy


ref = [t.clone() for t in step(engs[0])]
torch.cuda.synchronize()
names = ['down-scaled frames', 'thresholded maps', 'filtered maps', 'centres']
first = {n: 0 for n in names}
bad = 0
for it in range(ITERS):
    outs = []
    for i in range(N):
        with torch.cuda.stream(sts[i]):
            outs.append(step(engs[i]))
    torch.cuda.synchronize()
    for i in range(N):
        eq = [torch.equal(a, b) if a.dtype != torch.float64 else bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all()) for a, b in zip(outs[i], ref)]
        if not all(eq):
            bad += 1
            k = eq.index(False)
            first[names[k]] += 1
            if bad <= 6:
                print('iter %d engine %d: first differing stage: %s  (%s)' % (it, i, names[k], ', '.join('%s %s' % (n, 'same' if e else 'DIFFERS') for n, e in zip(names, eq))), flush=True)
print('%d steps, %d differ from the reference; first differing stage: %s' % (ITERS * N, bad, first))
