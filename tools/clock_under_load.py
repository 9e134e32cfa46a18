"""The shader clock the chip holds: idle, with one network pass running and with N passes sharing the chip (GPU box helper; tools/micro/clock_probe.hip built as tools/micro/libclock_probe.so).  argv: [N]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, scheduler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
WARM = float(os.environ.get('WARM', '2.5'))                  # seconds of load in front of every probe
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'libclock_probe.so'))
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
dev = torch.device('cuda', 0)
fr = torch.from_numpy(synth.blob_frames(32, 140, 250, seed=0)).cuda()
sts = scheduler.lane_streams(dev, N)
probe_stream = torch.cuda.Stream()
engs = [ops.Engine(seed=0) for _ in range(N)]
outs = [torch.empty((32, 140, 250), dtype=torch.uint8, device='cuda') for _ in range(N)]
res = torch.zeros(2, dtype=torch.int64, device='cuda')


def probe(n, ms=20.0):
    """clock during `ms` of wall time while n streams run network passes back to back"""
    torch.cuda.synchronize()
    passes = int(ms / 1.1 * max(n, 1)) // max(n, 1) + 4
    t0 = time.perf_counter()
    while n and time.perf_counter() - t0 < WARM:              # load first (the clock follows the load with a delay)
        for i in range(n):
            with torch.cuda.stream(sts[i]): engs[i].saliency(fr, out=outs[i])
        if (time.perf_counter() - t0) * 1e3 % 50 < 1: torch.cuda.synchronize()
    lib.clock_probe_launch(ctypes.c_void_p(probe_stream.cuda_stream), ctypes.c_void_p(res.data_ptr()), ctypes.c_double(ms))
    t = time.perf_counter()
    for _ in range(passes):
        for i in range(n):
            with torch.cuda.stream(sts[i]): engs[i].saliency(fr, out=outs[i])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) * 1e3
    c, r = [int(v) for v in res.cpu()]
    return c / r * 100.0, wall / (passes * max(n, 1)) if n else 0.0


for rep in range(2):
    print('idle: %.0f MHz' % probe(0)[0])
    for n in (1, 2, N):
        mhz, ms = probe(n)
        print('%d stream(s) of network passes: %.0f MHz  (%.3f ms per pass)' % (n, mhz, ms))
