"""N>1 path on CPUs: world_size-2 gloo run of the video sharding + box all_gather."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from retargetvid_amd import dist as D

COUNTS = [230, 613, 1283, 450, 300, 299, 1000]


def _boxes(i):
    rng = np.random.RandomState(100 + i)
    return rng.randint(0, 640, (COUNTS[i], 4)).astype(np.int32)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    shards = D.shard_videos(COUNTS, world)
    local = {i: _boxes(i) for i in shards[rank]}
    allb = D.gather_boxes(local, COUNTS)
    ok = sorted(allb) == list(range(len(COUNTS))) and all(np.array_equal(allb[i], _boxes(i)) for i in allb)
    q.put((rank, ok, shards[rank]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_balanced_partition():
    for world in (1, 2, 4, 8):
        s = D.shard_videos(COUNTS, world)
        assert sorted(i for r in s for i in r) == list(range(len(COUNTS)))
        loads = [sum(COUNTS[i] for i in r) for r in s]
        assert max(loads) - min(loads) <= max(COUNTS)


def test_gather_boxes_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sorted(i for _, _, s in res for i in s) == list(range(len(COUNTS)))


def test_single_process_passthrough():
    local = {0: _boxes(0)}
    assert np.array_equal(D.gather_boxes(local, COUNTS[:1])[0], _boxes(0))
