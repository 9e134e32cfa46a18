#!/usr/bin/env python3
"""BASELINE config 3: a RetargetVid-shaped run — 200 synthetic 640x360 videos with the real
frame counts (read from the annotation fixtures), targets 1:3 and 3:1 (saliency once, boxes
twice), videos sharded over the ranks, boxes all_gathered, rank 0 writes
results/<run>/<vid>_<w>-<h>.txt (+ _info.txt) and scores them with the evaluator counterpart.

  python tools/run_config3.py [--videos 200] [--max-frames 0] [--out gpurun_out/config3]
  python -m torch.distributed.run --nproc-per-node N tools/run_config3.py ...

The IoU numbers are meaningless (synthetic pixels against human annotations); the run checks
plumbing, sharding determinism and throughput."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from retargetvid_amd import dist as D, evaluate as E, ops, smartVidCrop as S, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--videos', type=int, default=200)
    ap.add_argument('--max-frames', type=int, default=0, help='truncate every video (0 = real length)')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'config3'))
    ap.add_argument('--workers', type=int, default=8, help='videos in flight per GPU (S.crop_videos)')
    ap.add_argument('--stream-batch', type=int, default=int(os.environ.get('STREAM_BATCH', 64)), help='maps per tail call inside the ingest (pipeline.StreamPipeline); 0 = one call per video')
    ap.add_argument('--packed', type=int, default=1, help='1: the job-level scheduler (retargetvid_amd/scheduler.py: full network chunks across video boundaries); 0: one video per worker thread (round 3)')
    ap.add_argument('--resident', type=int, default=1, help='1: the frames the job selects are generated before the timed run and lie in HBM (synth.ResidentBlobVideo), as bench.py\'s batch does; 0: generated on the fly inside the run (40 element-wise passes per frame on the GPU: the generator, not the path)')
    ap.add_argument('--repeat', type=int, default=1, help='run the job this many times in the process (the first pays one-time costs)')
    ap.add_argument('--ranks-per-gpu', type=int, default=1, help='processes that share one GPU (launch nproc-per-node = GPUs x this)')
    ap.add_argument('--annotations', default=os.path.join(ROOT, 'tests', 'golden', 'retargetvid'))
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    rpg = max(1, args.ranks_per_gpu)
    torch.cuda.set_device(local // rpg)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if rpg > 1:       # several ranks share a GPU (the host side of a video is Python: one interpreter lock per process); RCCL wants
            torch.distributed.init_process_group('gloo')     # one rank per device, so the boxes are gathered over gloo here
        else:
            torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))
    annots = E.load_annotations(args.annotations)
    vids = E.VID_INDS[:args.videos]
    counts = [len(annots[0]['1-3'][v]) for v in vids]
    if args.max_frames:
        counts = [min(c, args.max_frames) for c in counts]
    CP = S.sc_init_crop_params()
    ratios = ('1:3', '3:1')

    def make(i):
        def build():
            n = counts[i]
            cuts = synth.retargetvid_cuts(vids[i], n)[:-1]
            frames = resident[i] if args.resident else synth.LazyBlobVideo(n, seed=vids[i])
            return dict(fr=30.0, frame_count=n, w=640, h=360, frames=frames, trans_inds=cuts + [n])
        return build

    resident = {}
    if args.resident:                                        # which videos are this rank's is the job's decision: generate for all it may take
        mine = D.shard_videos(counts, world)[rank]
        for i in mine:
            cuts = synth.retargetvid_cuts(vids[i], counts[i])[:-1]
            sel = S._select_frames(counts[i], counts[i], cuts + [counts[i]], CP['skip'], CP['read_batch'])[0]
            resident[i] = synth.ResidentBlobVideo(counts[i], sel, seed=vids[i])
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    sched_stats = {}
    crop_fn = lambda vs, cp, rs, w: S.crop_videos(vs, cp, rs, workers=w, stream_batch=args.stream_batch, packed=bool(args.packed),
                                                  stats=sched_stats)
    runs = []
    for _ in range(max(1, args.repeat)):
        allb, st = D.crop_job(make, counts, ['%03d' % v for v in vids], CP, ratios, out_dir=args.out, workers=args.workers,
                              run_name='synthetic_default', crop_fn=crop_fn)
        runs.append(round(st['seconds_rank'], 3))
    dt_max = st['seconds_rank']
    if world > 1:                                            # the job's compute time = the slowest rank's
        t_ = torch.tensor([dt_max], dtype=torch.float64, device='cuda' if torch.distributed.get_backend() == 'nccl' else 'cpu')
        torch.distributed.all_reduce(t_, op=torch.distributed.ReduceOp.MAX)
        dt_max = float(t_.item())
    if rank == 0:
        score = None
        if args.videos == 200 and not args.max_frames:
            rows, _ = E.evaluate(args.out, args.annotations, out_path=os.path.join(args.out, 'eval_current.txt'))
            score = {ar: [round(x, 3) for x in s] for ar, s in rows[0][1].items()}
        dt = st['seconds_rank']
        print(json.dumps(dict(config='RetargetVid-shaped synthetic set', packed=bool(args.packed), seconds_rank0_runs=runs, scheduler={k: (round(v, 4) if isinstance(v, float) else v) for k, v in sched_stats.items()}, resident=bool(args.resident), stream_batch=args.stream_batch, videos=len(vids), world=world,
                              video_frames=sum(counts), saliency_frames_rank0=st['saliency_frames_rank'],
                              seconds_rank0=round(dt, 2), seconds_slowest_rank=round(dt_max, 2), video_frames_per_s_job=round(sum(counts) / dt_max, 1), video_frames_per_s_rank0=round(st['video_frames_rank'] / dt, 1),
                              saliency_frames_per_s_rank0=round(st['saliency_frames_rank'] / dt, 1), eval=score)))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
