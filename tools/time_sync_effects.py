"""One network pass per host synchronisation: what does an idle device cost the next pass?  (GPU box helper)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth
eng = ops.Engine(seed=0)
fr = torch.from_numpy(synth.blob_frames(32, 140, 250, seed=0)).cuda()
out = torch.empty((32, 140, 250), dtype=torch.uint8, device='cuda')
st = torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def one(sleep):
    with torch.cuda.stream(st):
        tot = 0.0; wall = 0.0
        for _ in range(20):
            if sleep: time.sleep(sleep)
            t = time.perf_counter()
            e0.record(st); eng.saliency(fr, out=out); e1.record(st)
            st.synchronize()
            wall += time.perf_counter() - t
            tot += e0.elapsed_time(e1)
        return tot / 20, wall / 20 * 1e3
with torch.cuda.stream(st):
    for _ in range(5): eng.saliency(fr, out=out)
    st.synchronize()
for sl in (0, 0.002, 0.02):
    ev, wall = one(sl)
    print('sync after every pass, %4.0f ms idle before it: device span %.3f ms, host wall %.3f ms' % (sl * 1e3, ev, wall))
