"""Kernel-family knobs priced where the headline is measured: time per network pass ALONE and with N passes sharing the chip
(GPU box helper).  argv: N 'NAME=V[,NAME=V...]' ...   (every argument after N is one knob set; the default set runs first)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, scheduler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SETS = [''] + sys.argv[2:]
NF = 32
fr = torch.from_numpy(synth.blob_frames(NF, 140, 250, seed=0)).cuda()
sts = scheduler.lane_streams(torch.device('cuda', 0), N)
outs = [torch.empty((NF, 140, 250), dtype=torch.uint8, device='cuda') for _ in range(N)]


def measure(knobs):
    kv = dict(x.split('=') for x in knobs.split(',') if x)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    engs = [ops.Engine(seed=0) for _ in range(N)]
    for k, v in old.items():
        if v is None: os.environ.pop(k)
        else: os.environ[k] = v
    res = []
    for n in (1, N):
        def run(k):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(k):
                for i in range(n):
                    with torch.cuda.stream(sts[i]): engs[i].saliency(fr, out=outs[i])
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / (k * n) * 1e3
        run(3)
        res.append(min(run(20) for _ in range(3)))
    for e in engs: e.close()
    return res


base = None
for s in SETS:
    r = measure(s)
    if base is None: base = r
    print('%-40s alone %.3f ms (%+5.1f us)   shared x%d %.3f ms (%+5.1f us)' % (s or 'default', r[0], (r[0] - base[0]) * 1e3, N, r[1], (r[1] - base[1]) * 1e3))
