#!/usr/bin/env python3
"""The 200-video job (BASELINE config 3's shape, frames resident in HBM) through retargetvid_amd.scheduler.JobScheduler:
seconds per run and the feeder's own time split, for a scheduler kept across runs and for a fresh one per run.
  python tools/time_scheduler.py [--lanes 4] [--runs 6] [--fresh 0|1] [--videos 200]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from retargetvid_amd import evaluate as E, scheduler, smartVidCrop as S, synth, weights   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lanes', type=int, default=4)
    ap.add_argument('--runs', type=int, default=6)
    ap.add_argument('--fresh', type=int, default=0)
    ap.add_argument('--videos', type=int, default=200)
    ap.add_argument('--chunk', type=int, default=32)
    ap.add_argument('--depth', type=int, default=2)
    ap.add_argument('--host-threads', type=int, default=3)
    ap.add_argument('--best', type=int, default=0, help='1: the ISM 2021 parameter set (use_best_settings: clustering at 35x62, focus stability on the host)')
    args = ap.parse_args()
    torch.cuda.set_device(0)
    annots = E.load_annotations(os.path.join(ROOT, 'tests', 'golden', 'retargetvid'))
    vids = E.VID_INDS[:args.videos]
    counts = [len(annots[0]['1-3'][v]) for v in vids]
    CP = S.sc_init_crop_params(use_best_settings=bool(args.best))
    videos = []
    for i, n in enumerate(counts):
        rng = np.random.RandomState(vids[i])
        cuts = sorted(set([0] + [int(c) for c in rng.randint(20, max(21, n - 20), rng.randint(0, 4))]))
        sel = S._select_frames(n, n, cuts + [n], CP['skip'], CP['read_batch'])[0]
        videos.append(dict(fr=30.0, frame_count=n, w=640, h=360, frames=synth.ResidentBlobVideo(n, sel, seed=vids[i]), trans_inds=cuts + [n]))
    torch.cuda.synchronize()
    sd = weights.make_synthetic_state_dict(0)
    js = None
    rows = []
    for r in range(args.runs):
        t0 = time.perf_counter()
        if js is None or args.fresh:
            js = scheduler.JobScheduler(CP, ('1:3', '3:1'), lanes=args.lanes, chunk=args.chunk, state_dict=sd, depth=args.depth,
                                        host_threads=args.host_threads)
        t1 = time.perf_counter()
        out = js.run(videos)
        t2 = time.perf_counter()
        del out
        if args.fresh:
            js.close()
        t3 = time.perf_counter()
        rows.append(dict(create=round(t1 - t0, 3), run=round(t2 - t1, 3), close=round(t3 - t2, 3),
                         device_side=round(js.stats['seconds_device_side'], 3), host_drain=round(js.stats['seconds_host_stage_drain'], 3), feeder=js.stats['feeder_seconds'],
                         mem_GB=round(torch.cuda.memory_reserved() / 2**30, 2)))
    print(json.dumps(dict(best_settings=bool(args.best), lanes=args.lanes, chunk=args.chunk, depth=args.depth, fresh=bool(args.fresh), saliency_frames=js.stats['network_frames'] + len(videos),
                          runs=rows)))


if __name__ == '__main__':
    main()
