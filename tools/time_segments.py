"""What does each stage of the network pass cost -- alone and with four passes sharing the chip?  (GPU box helper)
A stage is priced by LEAVING IT OUT (SVC_SEG_OFF, a measurement aid of the library: the launches of the stages in the
bitmask are skipped, the maps are garbage): cost = time of the full pass - time without the stage, per pass, on 1 stream
and on N streams (default 4: the bench's batches in flight).  argv: [N]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, scheduler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NF = 32
STAGES = ['front', 'blocks 2-3', 'blocks 4-7', 'blocks 8-14', 'blocks 15-17', 'features.18 + skips + post_cnn', 'upsampling 1',
          'upsampling 2', 'adapt / smooth / quantise']
fr = torch.from_numpy(synth.blob_frames(NF, 140, 250, seed=0)).cuda()      # the maps' geometry: what the bench's passes see behind k_cv_resize
dev = torch.device('cuda', 0)
sts = scheduler.lane_streams(dev, N)
outs = [torch.empty((NF, 140, 250), dtype=torch.uint8, device='cuda') for _ in range(N)]


def measure(mask):
    os.environ['SVC_SEG_OFF'] = str(mask)
    engs = [ops.Engine(seed=0) for _ in range(N)]
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


full = measure(0)
print('full pass: %.3f ms alone, %.3f ms per pass with %d streams' % (full[0], full[1], N))
tot = [0.0, 0.0]
for i, name in enumerate(STAGES):
    r = measure(1 << i)
    c = (full[0] - r[0], full[1] - r[1])
    tot[0] += c[0]; tot[1] += c[1]
    print('%-34s alone %7.1f us   shared %7.1f us   ratio %.2f' % (name, c[0] * 1e3, c[1] * 1e3, c[1] / c[0] if c[0] > 0 else 0))
print('%-34s alone %7.1f us   shared %7.1f us' % ('sum of the stages', tot[0] * 1e3, tot[1] * 1e3))
