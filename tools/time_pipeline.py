"""Steady-state ms per 32-frame batch with P engines in flight: network only vs network + tail (GPU box helper)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, smartVidCrop as S
CP = S.sc_init_crop_params()
fr = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=100)).cuda()
flags = np.zeros(32, np.uint8); flags[:2] = 1
for P in (1, 2, 4, 6):
    engs = [ops.Engine(seed=0) for _ in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    for mode in ('net', 'net+tail', 'tail'):
        def step(i):
            with torch.cuda.stream(streams[i % P]):
                e = engs[i % P]
                if mode != 'tail':
                    small = e.resize_frames(fr, 140, 250); maps = e.saliency(small)
                    step.maps = maps
                else:
                    maps = step.maps.clone()
                if mode != 'net':
                    e.threshold_(maps, 120); e.cluster_center_(maps, flags, CP)
        step.maps = None
        if mode == 'tail':
            m = engs[0].saliency(engs[0].resize_frames(fr, 140, 250)); torch.cuda.synchronize(); step.maps = m
        for i in range(2 * P): step(i)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(60): step(i)
        torch.cuda.synchronize(); print('P=%d %-9s %.3f ms/step' % (P, mode, (time.perf_counter() - t) / 60 * 1e3), flush=True)
    for e in engs: e.close()
