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
