"""`python bench.py --gpus N` starts N ranks itself (torch.distributed.run as a child, before the parent touches
the GPU).  On a box with fewer than N GPUs every rank must say so and the launcher's failure must be the exit code."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_n_ranks():
    if torch.cuda.device_count() >= 2:
        return                              # a real multi-GPU node: the driver's SCALE run covers it
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode != 0
    assert out.count('--gpus 2 but only') == 2, out[-2000:]     # both ranks started and failed loudly


def test_bench_rejects_mismatched_world():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode != 0 and 'WORLD_SIZE=2' in (p.stdout + p.stderr)
