import cProfile, pstats, sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from retargetvid_amd import ops, smartVidCrop as S, synth
eng = ops.Engine(seed=0); CP = S.sc_init_crop_params()
def vid(k):
    n = 500 + 13 * k
    return dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.LazyBlobVideo(n, seed=k), trans_inds=[0, 100 + k, 300, n])
S.smart_vid_crop_ratios(vid(0), CP, ('1:3', '3:1'), engine=eng)
pr = cProfile.Profile(); pr.enable()
for k in range(1, 25): S.smart_vid_crop_ratios(vid(k), CP, ('1:3', '3:1'), engine=eng)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
