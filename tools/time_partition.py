"""Does a spatial split pay?  The pipelined step with the clustering tail on its own HIP stream restricted to T CUs
(hipExtStreamCreateWithCUMask) and the network streams restricted to the other 256 - T, against the product's layout (one
unmasked stream per batch in flight, network and tail in order on it).  The tail is one workgroup of 1 024 threads / 125 VGPRs
/ up to 146 KB of LDS per map: it needs an EMPTY CU, which it waits for while other streams' network kernels keep refilling
the CUs; on reserved CUs it never waits.  (GPU box helper.)  argv: layouts "P:T" = batches in flight : tail CUs (0 = product layout)."""
import ctypes, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, weights, smartVidCrop as S
hip = ctypes.CDLL('libamdhip64.so')
B = 32
frames = torch.from_numpy(synth.blob_frames(B, 360, 640, seed=100, n_blobs=2, sigma=(30, 44))).cuda()
CP = S.sc_init_crop_params()
sd = weights.make_synthetic_state_dict(0)
order = os.environ.get('MASK_ORDER', 'block')


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << (b - 32 * w) for b in bits if 32 * w <= b < 32 * w + 32) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def cu_sets(T):
    if order == 'block':
        return list(range(256 - T)), list(range(256 - T, 256))
    step = 256 // T                                     # every step-th CU belongs to the tail
    tail = [i * step for i in range(T)]
    return [c for c in range(256) if c not in set(tail)], tail


def run_layout(P, T, steps=120):
    flags = np.zeros(B, np.uint8)                       # one tail round per call (what the streaming schedule gives)
    engs = [ops.Engine(sd) for _ in range(P)]
    if T:
        net_bits, tail_bits = cu_sets(T)
        nets = [masked_stream(net_bits) for _ in range(P)]
        tails = [masked_stream(tail_bits) for _ in range(P)]
    elif T == -9:                                       # product layout, the tail replaced by a one-thread spin of the same duration (its footprint gone)
        nets = [torch.cuda.Stream() for _ in range(P)]
        tails = nets
    elif T == -3:                                       # one HIGH-PRIORITY stream per slot for network and tail alike (does priority change anything?)
        nets = [torch.cuda.Stream(priority=-1) for _ in range(P)]
        tails = nets
    elif T < 0:                                         # unmasked, but the tail on a stream of its own (T = -2: a high-priority one): the slot's next network pass runs beside it
        nets = [torch.cuda.Stream() for _ in range(P)]
        tails = [torch.cuda.Stream(priority=-1 if T == -2 else 0) for _ in range(P)]
    else:
        nets = [torch.cuda.Stream() for _ in range(P)]
        tails = nets
    maps = [[torch.empty((B, 140, 250), dtype=torch.uint8, device='cuda') for _ in range(2)] for _ in range(P)]
    xyh = [[torch.empty((B, 2), dtype=torch.float64).pin_memory() for _ in range(2)] for _ in range(P)]
    ev_net = [[torch.cuda.Event() for _ in range(2)] for _ in range(P)]
    ev_done = [[torch.cuda.Event() for _ in range(2)] for _ in range(P)]
    used = [[False, False] for _ in range(P)]

    def go(n):
        torch.cuda.synchronize(); t = time.perf_counter()
        for s in range(n):
            k, j = s % P, (s // P) & 1
            if used[k][j]:
                ev_done[k][j].synchronize()             # the slot's previous results have arrived
            with torch.cuda.stream(nets[k]):
                small = engs[k].resize_frames(frames, 140, 250)
                engs[k].saliency(small, out=maps[k][j])
                engs[k].threshold_(maps[k][j], CP['t_threshold'])
                ev_net[k][j].record(nets[k])
            with torch.cuda.stream(tails[k]):
                if tails[k] is not nets[k]:
                    tails[k].wait_event(ev_net[k][j])
                if T == -9:
                    torch.cuda._sleep(SPIN)
                else:
                    xy = engs[k].cluster_center_(maps[k][j], flags, CP)
                    xyh[k][j].copy_(xy, non_blocking=True)
                ev_done[k][j].record(tails[k])
            used[k][j] = True
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    go(3 * P)
    ms = sorted(go(steps) for _ in range(5))[2]
    for e in engs:
        e.close()
    return ms


# cycles of torch.cuda._sleep that last as long as one lone tail round of this batch (~0.5 ms)
def _calibrate():
    e = ops.Engine(sd)
    m = e.saliency(e.resize_frames(frames, 140, 250)); e.threshold_(m, CP['t_threshold'])
    fl = np.zeros(B, np.uint8)
    for _ in range(3): e.cluster_center_(m.clone(), fl, CP)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): e.cluster_center_(m.clone(), fl, CP)
    torch.cuda.synchronize(); tail_ms = (time.perf_counter() - t) / 10 * 1e3
    torch.cuda._sleep(1000000); torch.cuda.synchronize(); t = time.perf_counter()
    torch.cuda._sleep(10000000); torch.cuda.synchronize(); per = (time.perf_counter() - t) * 1e3 / 10000000
    e.close()
    return int(tail_ms / per), tail_ms
SPIN, TAIL_MS = _calibrate()
print('one lone tail round: %.3f ms = %d spin cycles' % (TAIL_MS, SPIN), flush=True)

for lay in sys.argv[1:] or ['4:0', '4:16', '4:32', '5:32', '6:32']:
    P, T = (int(v) for v in lay.split(':'))
    ms = run_layout(P, T)
    print('%d batches in flight, tail on %3d reserved CUs (%s): %.4f ms per step of %d frames = %.0f frames/s' % (P, max(T, 0), ('the tail replaced by a one-thread spin of its duration' if T == -9 else 'own unmasked stream' if T == -1 else 'own HIGH-PRIORITY stream' if T == -2 else 'one high-priority stream per slot') if T < 0 else (order if T else 'product layout'), ms, B, B / ms * 1e3), flush=True)
