"""-m gpu: the multi-rank code path of bench.py on RCCL with the one GPU a test box has: a child
`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with BENCH_FORCE_DIST=1 takes
init_process_group('nccl'), the barriers, the MAX all-reduce of the timings and the box gather (dist.gather_boxes)
on the device -- and runs the sharded 200-video job (config.config3: dist.crop_job over RCCL), whose gathered windows must be
those of a single process without torch.distributed.  This module touches no GPU before the child has finished (no `engine`
fixture): the child must be the first to initialise it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_takes_the_rccl_path_at_world_1():
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ, BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--cpu-sample', '0', '--repeats', '1', '--iso-steps', '1']
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['config']['world_size_seen_by_rccl'] == 1
    assert d['value'] > 0 and d['steps'] == 2
    c3 = d['config']['config3']
    assert c3['videos'] == 200 and c3['n_gpus'] == 1 and c3['per_rank_fixed_costs_s']['rccl_init'] > 0
    # the same job in THIS process, no process group, frames generated on the fly: the windows must be the gathered ones
    import numpy as np
    from retargetvid_amd import evaluate as E, smartVidCrop as S, synth
    fcs = E.frame_counts(os.path.join(ROOT, 'tests', 'golden', 'retargetvid'))
    vids = list(E.VID_INDS)
    videos = [dict(fr=30.0, frame_count=fcs[v], w=640, h=360, frames=synth.LazyBlobVideo(fcs[v], seed=v),
                   trans_inds=synth.retargetvid_cuts(v, fcs[v])) for v in vids]
    res = S.crop_videos(videos, S.sc_init_crop_params(), ('1:3', '3:1'), workers=4)
    boxes = {r: {i: np.asarray(res[i][r][0]['bbs_np'], np.int32) for i in range(len(vids))} for r in ('1:3', '3:1')}
    assert synth.windows_crc32(boxes, ('1:3', '3:1'), len(vids)) == c3['windows_crc32']
    assert sum(len(b) for b in boxes['1:3'].values()) == c3['video_frames'] == 122684


def test_bench_self_launch_at_world_2_with_the_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2` as a user types it: the self-launch (a child torch.distributed.run, nothing exec'ed), two ranks,
    the sharding of the batches and of the 200-video job by rank, the barriers, the MAX over ranks, the gather of the boxes.  A test
    box has ONE GPU and RCCL refuses two ranks on one device, so BENCH_SHARE_GPU=1 puts both ranks on it and the (<= 16 B per
    frame) exchange on gloo: the numbers mean nothing, the path is the one `--gpus N` takes on a multi-GPU node, and the gathered
    windows of the job must be those of one process."""
    env = dict(os.environ, BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0', BENCH_VARIANT='0', BENCH_TORCH_BASELINE='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--cpu-sample', '0',
           '--repeats', '1', '--iso-steps', '1']
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                   # rank 0 prints, rank 1 does not
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 4
    cfg = d['config']
    assert cfg['world_size_seen_by_rccl'] == 2 and len(cfg['per_rank_frames_per_s']) == 2 and cfg['ranks_share_gpus']
    assert cfg['parallelism'].endswith('dp2')
    # value = the frames of BOTH ranks over the slowest rank's time
    assert abs(d['value'] - 2 * 32 * 4 / (d['ms_per_step'] * 4e-3)) < 1e-3 * d['value']
    c3 = cfg['config3']
    assert c3['videos'] == 200 and c3['n_gpus'] == 2 and c3['video_frames'] == 122684
    # the windows two ranks computed and gathered are those of one process (the checksum the world-1 test above derives in-process
    # from crop_videos; the frames, cuts and weights are functions of the seeds only)
    assert c3['windows_crc32'] == 508678473
    assert cfg['config3_host_fed'] is None and cfg['config3_shot_net'] is None      # N = 1 only
