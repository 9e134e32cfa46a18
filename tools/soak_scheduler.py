#!/usr/bin/env python3
"""Soak run of the job-level scheduler: random jobs (videos of 7 ... 400 frames, cuts anywhere incl. next to each other and
next to the ends, three geometries mixed, host arrays and on-device generators, several read batches) on random lane counts
and SMALL lane storages (drains and replacements in mid-job), both parameter sets -- every video's windows, centres and
filtered maps must equal smart_vid_crop_ratios on that video alone.  python tools/soak_scheduler.py [jobs] [seed]  (GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, scheduler, smartVidCrop as S, synth, weights
jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
sd = weights.make_synthetic_state_dict(0)
eng = ops.Engine(sd)
bad = n_vid = 0
for job in range(jobs):
    best = bool(job % 2)
    CP = dict(S.sc_init_crop_params(use_best_settings=best), read_batch=int(rng.choice([2000, 2000, 90, 40])), skip=int(rng.choice([6, 6, 4, 9])))
    vids = []
    for k in range(rng.randint(3, 14)):
        n = int(rng.choice([rng.randint(7, 30), rng.randint(30, 120), rng.randint(120, 400)]))
        h, w = [(360, 640), (360, 640), (480, 640), (640, 360)][rng.randint(0, 4)]
        cuts = sorted(set([0] + [int(c) for c in rng.randint(1, max(2, n - 2), rng.randint(0, 5))]))
        if rng.rand() < 0.3 and n > 12:
            cuts = sorted(set(cuts + [cuts[-1] + 1, cuts[-1] + 2]))          # cuts next to each other
        cuts = [c for c in cuts if n - c >= 2]
        lazy = rng.rand() < 0.4
        frames = synth.LazyBlobVideo(n, h, w, seed=int(rng.randint(0, 10**6))) if lazy else synth.blob_frames(n, h, w, seed=int(rng.randint(0, 10**6)))
        vids.append(dict(fr=float(rng.choice([25.0, 30.0, 29.97])), frame_count=n, w=w, h=h, frames=frames, trans_inds=cuts + [n]))
    seq = [S.smart_vid_crop_ratios(v, CP, ('1:3', '3:1'), engine=eng) for v in vids]
    lanes, rows = int(rng.randint(1, 7)), int(rng.choice([24, 64, 200, 4096]))
    js = scheduler.JobScheduler(CP, ('1:3', '3:1'), lanes=lanes, state_dict=sd, lane_rows=rows)
    try:
        par = js.run(vids)
    finally:
        js.close()
    ok = True
    for a, b in zip(seq, par):
        for r in ('1:3', '3:1'):
            ok = ok and a[r][0]['bbs'] == b[r][0]['bbs'] and a[r][0]['dx'] == b[r][0]['dx'] and a[r][0]['dy'] == b[r][0]['dy']
        ok = ok and torch.equal(a['1:3'][0]['smaps_dev'], b['1:3'][0]['smaps_dev'])
    n_vid += len(vids)
    bad += 0 if ok else 1
    print('job %2d: %2d videos, %d lanes, %4d rows per lane, best=%d, read_batch %4d, skip %d, chunks %3d (%.0f %% full): %s' % (
        job, len(vids), lanes, rows, best, CP['read_batch'], CP['skip'], js.stats['chunks'], 100 * js.stats['mean_chunk_fill'], 'identical' if ok else 'MISMATCH'), flush=True)
print('%d jobs, %d videos, %d mismatching jobs' % (jobs, n_vid, bad))
sys.exit(1 if bad else 0)
