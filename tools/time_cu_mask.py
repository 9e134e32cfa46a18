"""Network passes on N streams, every stream restricted to a share of the CUs (hipExtStreamCreateWithCUMask): does spatial
partitioning beat time sharing?  (GPU box helper)  argv: layouts, e.g. "4x64 4x256 2x128 4x96 8x32" = streams x CUs-per-stream;
CUs-per-stream 256 = unmasked.  MASK_ORDER=block (default: consecutive mask bits) | stride (every (256/cus)-th bit)."""
import ctypes, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth
hip = ctypes.CDLL('libamdhip64.so')
NF = int(os.environ.get('FRAMES', '32'))
fr = torch.from_numpy(synth.blob_frames(NF, 140, 250, seed=0)).cuda()
order = os.environ.get('MASK_ORDER', 'block')


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << (b - 32 * w) for b in bits if 32 * w <= b < 32 * w + 32) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


for lay in sys.argv[1:] or ['4x256', '4x64', '2x128', '4x128']:
    n, cus = (int(v) for v in lay.split('x'))
    sts = []
    for i in range(n):
        if cus >= 256:
            sts.append(torch.cuda.Stream())
        else:
            if order == 'block':
                first = (i * cus) % 256
                bits = [(first + j) % 256 for j in range(cus)]
            else:
                step = 256 // cus
                bits = [(i % step) + step * j for j in range(cus)]
            sts.append(masked_stream(bits))
    engs = [ops.Engine(seed=0) for _ in range(n)]
    outs = [torch.empty((NF, 140, 250), dtype=torch.uint8, device='cuda') for _ in engs]

    def run(k):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(k):
            for i in range(n):
                with torch.cuda.stream(sts[i]): engs[i].saliency(fr, out=outs[i])
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / (k * n) * 1e3
    run(3)
    print('%s (%s): %.3f ms per pass of %d frames' % (lay, order, run(15), NF), flush=True)
    for e in engs: e.close()
