import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from retargetvid_amd import ops
torch.cuda.init(); torch.zeros(1, device='cuda')
for i in range(3):
    t = time.perf_counter(); e = ops.Engine(seed=0); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
    print('Engine(seed=0): create %.1f ms, close %.1f ms' % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
from retargetvid_amd import weights
t = time.perf_counter(); sd = weights.make_synthetic_state_dict(0); print('make_synthetic_state_dict %.1f ms' % ((time.perf_counter() - t) * 1e3))
for i in range(2):
    t = time.perf_counter(); e = ops.Engine(sd); t1 = time.perf_counter(); print('Engine(sd): create %.1f ms' % ((t1 - t) * 1e3)); e.close()
