"""Is the multi-video job WITH shot detection inside reproducible run after run?  NV RetargetVid-shaped videos (every frame resident, no
trans_inds), JobScheduler with a TransNet (synthetic weights) planning ahead of the lanes in its planner threads, R runs: segmentations,
selected frames and crop windows of every video must equal the first run's.   python tools/soak_shot_job_repeat.py [NV] [R]   (GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import evaluate as E, scheduler, smartVidCrop as S, synth, weights, transnetv1_handler as TN
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 40
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fcs = E.frame_counts(os.path.join(ROOT, 'tests', 'golden', 'retargetvid'))
vids = list(E.VID_INDS)[:NV]
dev = torch.device('cuda', 0)
videos = []
for v in vids:
    n = fcs[v]
    src = synth.LazyBlobVideo(n, 360, 640, seed=v, device=dev)
    fr = torch.cat([src.select(range(a, min(n, a + 64))) for a in range(0, n, 64)])
    videos.append(dict(fr=30.0, frame_count=n, w=640, h=360, frames=fr))
tsd = weights.make_transnet_state_dict(0)                      # unbiased synthetic weights would report impossible segmentations: bias to few cuts
tsd['TransNet/dense_1/kernel'] = tsd['TransNet/dense_1/kernel'] * np.float32(0.02)
tsd['TransNet/dense_1/bias'] = np.array([12.0, -12.0], np.float32)
net = TN.ShotTransNet(TN.ShotTransNetParams(), weights=tsd)
CP = S.sc_init_crop_params()
js = scheduler.JobScheduler(CP, ('1:3', '3:1'), lanes=int(os.environ.get('LANES', 12)), state_dict=weights.make_synthetic_state_dict(0), shot_net=net)


def digest(out):
    return [(tuple(o[r][0]['true_inds']), np.asarray(o[r][0]['segmentation']).tobytes(), tuple(map(tuple, o[r][0]['bbs']))) for o in out for r in ('1:3', '3:1')]


ref = digest(js.run(videos))
bad = 0
for k in range(1, R):
    d = digest(js.run(videos))
    diff = [i // 2 for i, (a, b) in enumerate(zip(ref, d)) if a != b]
    if diff:
        bad += 1
        print('run %d: videos %s differ from the first run' % (k, sorted(set(diff))[:8]), flush=True)
print('%d videos, pipe %s, %d planner threads: %d of %d runs differ from the first run' % (NV, net.matrix_pipe(), len(js._plan_nets), bad, R - 1))
js.close(); net.close()
