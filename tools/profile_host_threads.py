"""cProfile of the worker threads of smartVidCrop.crop_videos (config 3's shape: videos in flight on one GPU).  Every worker
profiles its own smart_vid_crop_ratios calls (wall clock, so time spent waiting for the GIL or the GPU shows where it is
waited for); the tables are merged.  python tools/profile_host_threads.py [videos] [workers] [stream_batch]"""
import cProfile, pstats, sys, os, threading, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from retargetvid_amd import smartVidCrop as S, synth
nv, workers, sb = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 60), (2, 4), (3, 64)))
CP = S.sc_init_crop_params()
def vid(k):
    n = 400 + 37 * (k % 11)
    return lambda: dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.LazyBlobVideo(n, seed=k), trans_inds=[0, 100 + k, 300, n])
S.crop_videos([vid(k) for k in range(8)], CP, ('1:3', '3:1'), workers=workers, stream_batch=sb)      # warm-up
profs, lock = [], threading.Lock()
orig = S.smart_vid_crop_ratios
def wrapped(*a, **kw):
    pr = cProfile.Profile()
    pr.enable()
    try:
        return orig(*a, **kw)
    finally:
        pr.disable()
        with lock: profs.append(pr)
S.smart_vid_crop_ratios = wrapped
t = time.perf_counter()
S.crop_videos([vid(k) for k in range(nv)], CP, ('1:3', '3:1'), workers=workers, stream_batch=sb)
dt = time.perf_counter() - t
print('%d videos, %d workers, stream_batch %d: %.3f s = %.2f ms per video' % (nv, workers, sb, dt, dt / nv * 1e3))
st = pstats.Stats(profs[0])
for p in profs[1:]: st.add(p)
st.sort_stats('tottime').print_stats(32)
