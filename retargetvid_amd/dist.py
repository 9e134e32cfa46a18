"""Multi-GPU sharding of the saliency-to-crop path (one process per GPU, torch.distributed).

The reference is single-process, single-GPU, batch 1 ("no collectives").
The path shards by video: every video is independent (the only cross-frame coupling,
the cut blend, stays inside a video), so ranks process disjoint videos with no
data-path collective, and one all_gather of the final int32 boxes per job makes every
rank (in particular rank 0, which writes the result files) hold all crop windows.
On GPUs the backend is "nccl" (= RCCL over xGMI); the payload is <= 16 B per frame, so
the exchange is latency-bound.  The same code runs under "gloo" on CPUs for tests."""
import numpy as np
import torch
import torch.distributed as dist


def shard_videos(frame_counts, world_size):
    """Longest-processing-time-first assignment of videos to ranks.
    frame_counts: list of ints -> list (per rank) of video indices, deterministic."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-frame_counts[i], i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += frame_counts[i]
    return [sorted(v) for v in out]


def gather_boxes(local, frame_counts, device=None):
    """local: {video_index: int array [fc,4]} computed by this rank.
    -> {video_index: np.int32 [fc,4]} for ALL videos on every rank (one all_gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: np.asarray(v, np.int32) for k, v in local.items()}
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = shard_videos(frame_counts, world)
    sizes = [sum(frame_counts[i] for i in s) for s in shards]
    cap = max(max(sizes), 1)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    buf = torch.zeros((cap, 4), dtype=torch.int32, device=device)
    if shards[rank]:
        mine = np.concatenate([np.asarray(local[i], np.int32).reshape(-1, 4) for i in shards[rank]])
        buf[:mine.shape[0]] = torch.from_numpy(mine).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    result = {}
    for r in range(world):
        arr = out[r].cpu().numpy()
        pos = 0
        for i in shards[r]:
            result[i] = arr[pos:pos + frame_counts[i]].copy()
            pos += frame_counts[i]
    return result


def crop_job(make_video, frame_counts, names, CP, ratios, out_dir=None, workers=4, crop_fn=None, run_name='run'):
    """A whole multi-video job on however many ranks there are (BASELINE config 3's shape): videos are
    sharded over the ranks by frame count, every rank crops its share (``crop_fn``, default
    smartVidCrop.crop_videos on this rank's GPU), ONE all_gather per target ratio makes every rank hold all
    crop windows, the per-video info dicts are gathered to rank 0, and rank 0 writes
    ``<out_dir>/<run_name>/<name>_<w>-<h>.txt`` (+ ``_info.txt``) in the reference's result format
    (smartVidCrop.py:2730-2731, :2777-2785).

    make_video(i) -> zero-argument callable or ingest_pickle dict of video i; names[i] = file stem ('%03d' id).
    crop_fn(videos, CP, ratios, workers) -> list of {ratio: (VD, results)}; tests pass a CPU stand-in.
    Returns (boxes {ratio: {i: int32[fc,4]}} on every rank, stats dict)."""
    import time
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if crop_fn is None:
        from . import smartVidCrop as S
        crop_fn = lambda vids, cp, rs, w: S.crop_videos(vids, cp, rs, workers=w)
    ratios = tuple(ratios)
    mine = shard_videos(list(frame_counts), world)[rank]
    t0 = time.perf_counter()
    res_all = crop_fn([make_video(i) for i in mine], CP, ratios, workers) if mine else []
    dt = time.perf_counter() - t0
    local = {r: {} for r in ratios}
    infos, n_sal = {}, 0
    for i, res in zip(mine, res_all):
        for r in ratios:
            b = np.asarray(res[r][0]['bbs'], np.int32).reshape(-1, 4)
            if b.shape[0] != frame_counts[i]:
                raise ValueError('video %s: %d boxes for %d frames' % (names[i], b.shape[0], frame_counts[i]))
            local[r][i] = b
        infos[i] = {r: res[r][1] for r in ratios}
        n_sal += int(res[ratios[0]][0].get('fc_sel', 0))
    allb = {r: gather_boxes(local[r], list(frame_counts)) for r in ratios}      # the path's one exchange
    if world > 1:
        bucket = [None] * world if rank == 0 else None
        dist.gather_object(infos, bucket, dst=0)                                # host-side text, outside the data path
        if rank == 0:
            infos = {k: v for d in bucket for k, v in d.items()}
    if rank == 0 and out_dir is not None:
        import os
        from . import smartVidCrop as S
        run_dir = os.path.join(out_dir, run_name)
        for i, name in enumerate(names):
            for r in ratios:
                S.write_results(run_dir, name, r, {'bbs': allb[r][i].tolist()}, infos.get(i, {}).get(r, {}))
    stats = dict(world=world, rank=rank, videos_rank=len(mine), video_frames_rank=int(sum(frame_counts[i] for i in mine)),
                 saliency_frames_rank=n_sal, seconds_rank=dt)
    return allb, stats
