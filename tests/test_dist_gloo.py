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


# ---- the whole multi-video job (BASELINE config 3's shape) at world 2 on CPUs ------------------------------
# crop_fn stand-in: the oracle pipeline (CPU restatement of the same path) plays the GPU; sharding, the
# all_gather of the boxes, the info gather and rank 0's result files are the product's own code (dist.crop_job).
JOB_COUNTS = [31, 44, 26, 38, 29]
JOB_NAMES = ['%03d' % v for v in (1, 2, 3, 601, 602)]


def _job_video(i):
    from retargetvid_amd import synth
    n = JOB_COUNTS[i]
    return dict(fr=30.0, frame_count=n, w=160, h=90, frames=synth.blob_frames(n, 90, 160, seed=50 + i),
                trans_inds=[0, 12 + i, n])


def _oracle_crop_fn(videos, CP, ratios, workers):
    from oracle import pipeline_ref as P
    from retargetvid_amd import weights
    sd = weights.make_synthetic_state_dict(0)
    out = []
    for v in videos:
        v = v() if callable(v) else v
        per = {}
        for r in ratios:
            VD = P.smart_vid_crop(v, dict(P.init_crop_params(), out_ratio=r, hdbscan_min=5), sd)
            per[r] = (VD, {'result': 'smart cropped', 'cuts_clust': 0, 't_total': '  0.010s,  1.000%'})
        out.append(per)
    return out


def _job_worker(rank, world, port, out_dir, q):
    torch.set_num_threads(2)
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    allb, st = D.crop_job(lambda i: (lambda: _job_video(i)), JOB_COUNTS, JOB_NAMES, {}, ('1:3', '3:1'), out_dir=out_dir,
                          workers=1, crop_fn=_oracle_crop_fn, run_name='oracle_standin')
    q.put((rank, st['videos_rank'], {r: {i: allb[r][i].tolist() for i in allb[r]} for r in allb}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_crop_job_world2_equals_world1(tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    d1, d2 = str(tmp_path / 'w1'), str(tmp_path / 'w2')
    procs = [ctx.Process(target=_job_worker, args=(r, 2, port, d2, q)) for r in range(2)]
    procs.append(ctx.Process(target=_job_worker, args=(0, 1, 0, d1, q)))
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    boxes = [b for _, _, b in res]
    assert boxes[0] == boxes[1] == boxes[2]                       # every rank of both jobs holds all crop windows
    assert sorted(n for _, n, _ in res) in ([2, 3, 5], [1, 4, 5])
    run1, run2 = os.path.join(d1, 'oracle_standin'), os.path.join(d2, 'oracle_standin')
    files = sorted(os.listdir(run1))
    assert files == sorted(os.listdir(run2)) and len(files) == 4 * len(JOB_NAMES)
    for f in files:
        assert open(os.path.join(run1, f)).read() == open(os.path.join(run2, f)).read(), f
    rows = open(os.path.join(run2, '601_1-3.txt')).read().splitlines()
    assert len(rows) == JOB_COUNTS[3] and all(len(r.split(',')) == 4 for r in rows)
    assert 't_total' in open(os.path.join(run2, '601_1-3_info.txt')).read()   # info of a video rank 1 may have owned


# ---- resume unit (smartVidCrop.py:2732-2742) and a failure on one rank, world 2 -------------------------------------
def _counting_crop_fn(videos, CP, ratios, workers):
    """Stand-in that records which videos it was asked for (by frame count) and returns fixed windows."""
    out = []
    for v in videos:
        v = v() if callable(v) else v
        n = int(v['frame_count'])
        if CP.get('fail_on') == n:
            raise ValueError('synthetic failure on the video with %d frames' % n)
        bbs = [[j % 7, 0, 120 + j % 7, 360] for j in range(n)]
        out.append({r: ({'bbs': bbs, 'fc_sel': n // 6}, {'result': 'smart cropped', 'frames': n}) for r in ratios})
    return out


def _resume_worker(rank, world, port, out_dir, CP, replace, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        allb, st = D.crop_job(lambda i: dict(frame_count=JOB_COUNTS[i]), JOB_COUNTS, JOB_NAMES, CP, ('1:3', '3:1'),
                              out_dir=out_dir, workers=1, crop_fn=_counting_crop_fn, run_name='r', replace_existing=replace)
        q.put((rank, 'ok', st['videos_rank'], st['videos_skipped'], sorted(allb['1:3']), int(allb['3:1'][3][5][0])))
    except Exception as e:
        q.put((rank, 'error', type(e).__name__, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(out_dir, CP, replace):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + (0 if replace else 7)
    procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, out_dir, CP, replace, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_crop_job_skips_videos_whose_result_files_exist(tmp_path):
    out = str(tmp_path)
    first = _run_world2(out, {}, True)
    assert [r[1] for r in first] == ['ok', 'ok'] and sum(r[2] for r in first) == 5 and first[0][3] == 0
    run = os.path.join(out, 'r')
    os.remove(os.path.join(run, '601_3-1.txt'))                       # video 601 lost one of its files: it must be redone
    marker = os.path.join(run, '002_1-3.txt')
    open(marker, 'a').close()
    before = os.path.getmtime(marker)
    second = _run_world2(out, {}, False)
    assert [r[1] for r in second] == ['ok', 'ok']
    assert sum(r[2] for r in second) == 1 and second[0][3] == 4       # one video computed, four skipped
    assert second[0][4] == second[1][4] == [0, 1, 2, 3, 4]            # every rank still holds all windows
    assert second[0][5] == second[1][5] == 5
    assert os.path.getmtime(marker) == before                         # files of skipped videos are not rewritten
    assert len(open(os.path.join(run, '601_3-1.txt')).read().splitlines()) == JOB_COUNTS[3]


def test_crop_job_failure_on_one_rank_raises_on_every_rank(tmp_path):
    res = _run_world2(str(tmp_path), {'fail_on': JOB_COUNTS[1]}, True)
    assert [r[1] for r in res] == ['error', 'error']
    kinds = sorted(r[2] for r in res)
    assert kinds == ['RuntimeError', 'ValueError'], res               # the failing rank's own error, "another rank failed" on the other
    assert not os.path.isdir(os.path.join(str(tmp_path), 'r'))        # nothing was written
