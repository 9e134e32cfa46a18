"""Is the multi-video job reproducible run after run?  The same RetargetVid-shaped job (first NV videos, resident frames) through
JobScheduler with 4 and 12 lanes, R times each: the centres of every video must be bit-identical across runs and lane counts.
python tools/soak_job_repeat.py [NV] [R]   (GPU box; SVC_MX=f32 for the fp32 pipe)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import evaluate as E, scheduler, smartVidCrop as S, synth, weights
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 60
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fcs = E.frame_counts(os.path.join(ROOT, 'tests', 'golden', 'retargetvid'))
vids = list(E.VID_INDS)[:NV]
counts = [fcs[v] for v in vids]
CP = S.sc_init_crop_params()
sd = weights.make_synthetic_state_dict(0)
dev = torch.device('cuda', 0)
videos = []
for i, v in enumerate(vids):
    cuts = synth.retargetvid_cuts(v, counts[i])
    sel = S._select_frames(counts[i], counts[i], cuts, CP['skip'], CP['read_batch'])[0]
    videos.append(dict(fr=30.0, frame_count=counts[i], w=640, h=360, frames=synth.ResidentBlobVideo(counts[i], sel, seed=v, device=dev), trans_inds=cuts))
ref = None
for lanes in [int(v) for v in os.environ.get('LANES', '4,12,4,12').split(',')]:
    js = scheduler.JobScheduler(CP, ('1:3',), lanes=lanes, state_dict=sd)
    try:
        for r in range(R):
            out = js.run(videos)
            xy = [np.array([o['1:3'][0]['dx'], o['1:3'][0]['dy']]) for o in out]
            if ref is None:
                ref = xy
                print('reference: %d lanes, run 0' % lanes, flush=True)
                continue
            bad = [(i, int((a != b).any(0).sum()), float(np.abs(a - b).max())) for i, (a, b) in enumerate(zip(ref, xy)) if not np.array_equal(a, b)]
            if bad or R <= 8: print('%2d lanes run %d: %s' % (lanes, r, 'identical' if not bad else 'DIFFERS in %d videos: %s' % (len(bad), bad[:6])), flush=True)
            n_bad = globals().get('n_bad', 0) + (1 if bad else 0); n_all = globals().get('n_all', 0) + 1
    finally:
        js.close()

print('%d of %d runs differ from the reference run' % (globals().get('n_bad', 0), globals().get('n_all', 0)))
