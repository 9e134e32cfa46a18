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


def _agree_on_failure(err):
    """Every rank learns whether ANY rank failed before a collective (one MAX all-reduce of a flag), so that a failure
    on one rank raises everywhere instead of leaving the others waiting in the box gather."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if err is not None:
            raise err
        return
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if err is not None:
        raise err
    if int(flag.item()):
        raise RuntimeError('crop_job: another rank failed (see its traceback); no result files were written')


def result_paths(out_dir, run_name, name, ratio):
    import os
    stem = os.path.join(out_dir, run_name, name + '_' + str(ratio.replace(':', '-')))
    return stem + '.txt', stem + '_info.txt'


def read_boxes(path, frame_count):
    """A result file written by an earlier run -> int32 [fc,4] (None when it does not hold frame_count rows)."""
    try:
        b = np.loadtxt(path, dtype=np.int32, delimiter=',', ndmin=2)
    except Exception:
        return None
    if b.shape != (frame_count, 4) or (b[:, 0] > b[:, 2]).any() or (b[:, 1] > b[:, 3]).any() or (b < 0).any():
        return None                                            # not crop windows (x1 <= x2, y1 <= y2, inside the frame): computed again
    return b


def crop_job(make_video, frame_counts, names, CP, ratios, out_dir=None, workers=12, crop_fn=None, run_name='run',
             replace_existing=True):
    """A whole multi-video job on however many ranks there are (BASELINE config 3's shape): videos are
    sharded over the ranks by frame count, every rank crops its share (``crop_fn``, default
    smartVidCrop.crop_videos on this rank's GPU), ONE all_gather per target ratio makes every rank hold all
    crop windows, the per-video info dicts are gathered to rank 0, and rank 0 writes
    ``<out_dir>/<run_name>/<name>_<w>-<h>.txt`` (+ ``_info.txt``) in the reference's result format
    (smartVidCrop.py:2730-2731, :2777-2785).

    make_video(i) -> zero-argument callable or ingest_pickle dict of video i; names[i] = file stem ('%03d' id).
    crop_fn(videos, CP, ratios, workers) -> list of {ratio: (VD, results)}; tests pass a CPU stand-in.
    replace_existing=False: the reference driver's resume unit (smartVidCrop.py:2732-2742) -- a video whose
    ``.txt`` AND ``_info.txt`` exist (for every ratio, with one row per frame) is not computed again: its windows are
    read back from the files, which are left untouched.  Decided on rank 0 and broadcast, so that every rank shards
    the same remaining videos.
    Returns (boxes {ratio: {i: int32[fc,4]}} on every rank, stats dict)."""
    import time
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if crop_fn is None:
        from . import smartVidCrop as S
        crop_fn = lambda vids, cp, rs, w: S.crop_videos(vids, cp, rs, workers=w)
    import os
    ratios = tuple(ratios)
    n_all = len(frame_counts)
    kept = {}                                              # video -> {ratio: boxes read back from an earlier run}
    if not replace_existing and out_dir is not None:
        done, scan_err = [], None
        if rank == 0:
            try:                                               # a failure of the scan (an unreadable out_dir, a bad name) must not
                for i in range(n_all):                         # leave the other ranks waiting in the broadcast below
                    got = {}
                    for r in ratios:
                        txt, info = result_paths(out_dir, run_name, names[i], r)
                        b = read_boxes(txt, frame_counts[i]) if (os.path.isfile(txt) and os.path.isfile(info)) else None
                        if b is None:
                            break
                        got[r] = b
                    if len(got) == len(ratios):
                        kept[i] = got
                done = sorted(kept)
            except Exception as e:
                scan_err = e
        _agree_on_failure(scan_err)
        if world > 1:
            box = [done]
            dist.broadcast_object_list(box, src=0)
            done = box[0]
        skip = set(done)
    else:
        skip = set()
    todo = [i for i in range(n_all) if i not in skip]
    shards = shard_videos([frame_counts[i] for i in todo], world)
    mine = [todo[j] for j in shards[rank]]
    t0 = time.perf_counter()
    local = {r: {} for r in ratios}
    infos, n_sal, err = {}, 0, None
    try:
        res_all = crop_fn([make_video(i) for i in mine], CP, ratios, workers) if mine else []
        for i, res in zip(mine, res_all):
            for r in ratios:
                vd = res[r][0]
                b = np.asarray(vd['bbs_np'] if 'bbs_np' in vd else vd['bbs'], np.int32).reshape(-1, 4)
                if b.shape[0] != frame_counts[i]:
                    raise ValueError('video %s: %d boxes for %d frames' % (names[i], b.shape[0], frame_counts[i]))
                local[r][i] = b
            infos[i] = {r: res[r][1] for r in ratios}
            n_sal += int(res[ratios[0]][0].get('fc_sel', 0))
    except Exception as e:                                   # raised on every rank below, before any collective of the data path
        err = e
    dt = time.perf_counter() - t0
    _agree_on_failure(err)
    sub_counts = [frame_counts[i] for i in todo]
    allb = {}
    for r in ratios:                                         # the path's one exchange (per ratio), over the videos computed now
        got = gather_boxes({todo.index(i): b for i, b in local[r].items()}, sub_counts) if todo else {}
        allb[r] = {todo[j]: b for j, b in got.items()}
        if world > 1 and skip:                               # windows read back by rank 0 travel with the same call's result
            box = [{i: kept[i][r] for i in kept}] if rank == 0 else [None]
            dist.broadcast_object_list(box, src=0)
            allb[r].update(box[0])
        elif skip:
            allb[r].update({i: kept[i][r] for i in kept})
    if world > 1 and out_dir is not None:                    # (only the result files need the info dicts: no files, no object collective)
        bucket = [None] * world if rank == 0 else None
        dist.gather_object(infos, bucket, dst=0)                                # host-side text, outside the data path
        if rank == 0:
            infos = {k: v for d in bucket for k, v in d.items()}
    if rank == 0 and out_dir is not None:
        from . import smartVidCrop as S
        run_dir = os.path.join(out_dir, run_name)
        for i, name in enumerate(names):
            if i in skip:
                continue                                     # resume: files of an earlier run stay as they are
            for r in ratios:
                S.write_results(run_dir, name, r, {'bbs': allb[r][i].tolist()}, infos.get(i, {}).get(r, {}))
    stats = dict(world=world, rank=rank, videos_rank=len(mine), video_frames_rank=int(sum(frame_counts[i] for i in mine)),
                 saliency_frames_rank=n_sal, seconds_rank=dt, videos_skipped=len(skip))
    return allb, stats
