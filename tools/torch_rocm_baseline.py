"""How fast does the REFERENCE's way of running the network go on this MI355X?  (GPU box helper; a baseline, not product code.)

The reference runs UNISAL as eager PyTorch modules (cuDNN on its hardware; under PyTorch-ROCm the same calls go to MIOpen / rocBLAS):
3rd_party_libs/unisal/unisal/train.py:778-859 (run_inference_fast), model.py:411-506.  This script times exactly that formulation --
oracle/unisal_ref.forward_logits, the state-dict-driven restatement of the reference's modules (F.conv2d / F.batch_norm / F.interpolate,
BatchNorm NOT folded, NCHW fp32) -- on cuda:0 at the benchmark's shape (32 frames of 256 x 416 network input -> 140 x 250 maps, synthetic
weights), forward pass only: no LANCZOS, no u8 quantisation, no clustering tail (those run on the CPU in the reference).  Printed next to
this package's svc_saliency_u8 on the same frames, which includes LANCZOS + quantisation.

    python tools/torch_rocm_baseline.py [batch] [iters]

MIOpen compiles / searches kernels on first use: the first passes are warm-up (reported separately)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unisal_ref as U                     # noqa: E402  (tools/ may use the oracle; the product never does)
from retargetvid_amd import ops, synth, weights        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
sd = weights.make_synthetic_state_dict(0)
sd_dev = {k: (v if torch.is_tensor(v) else torch.as_tensor(v)).to(dev) for k, v in sd.items()}
_g = U.gaussian_maps
U.gaussian_maps = lambda g, h, w, scaling=6.0: _g(g.cpu(), h, w, scaling).to(dev)     # the priors are built on the host once per call, as model.py:348-378 does

x = torch.randn(B, 3, 256, 416, device=dev)


def once():
    return U.forward_logits(sd_dev, x, (140, 250))


t = time.perf_counter()
once(); torch.cuda.synchronize()
first = time.perf_counter() - t
t = time.perf_counter()
for _ in range(3):
    once()
torch.cuda.synchronize()
warm = (time.perf_counter() - t) / 3
times = []
for _ in range(ITERS):
    torch.cuda.synchronize(); t = time.perf_counter()
    once()
    torch.cuda.synchronize(); times.append(time.perf_counter() - t)
times.sort()
med = times[len(times) // 2]
print('PyTorch-ROCm eager (the reference\'s formulation), fp32 NCHW, batch %d x 256 x 416: first pass %.1f s, next three %.1f ms each, '
      'then median %.2f ms per pass (min %.2f) = %.0f frames/s  [torch %s, MIOpen benchmark=%s]'
      % (B, first, warm * 1e3, med * 1e3, times[0] * 1e3, B / med, torch.__version__, torch.backends.cudnn.benchmark))
torch.backends.cudnn.benchmark = True
for _ in range(3):
    once()
torch.cuda.synchronize()
times = []
for _ in range(ITERS):
    torch.cuda.synchronize(); t = time.perf_counter()
    once()
    torch.cuda.synchronize(); times.append(time.perf_counter() - t)
times.sort()
print('  with torch.backends.cudnn.benchmark = True (MIOpen picks per-shape kernels): median %.2f ms (min %.2f) = %.0f frames/s'
      % (times[len(times) // 2] * 1e3, times[0] * 1e3, B / times[len(times) // 2]))

eng = ops.Engine(sd, device=0)
fr = torch.from_numpy(synth.blob_frames(B, 140, 250, seed=0)).to(dev)
out = torch.empty((B, 140, 250), dtype=torch.uint8, device=dev)
for _ in range(3):
    eng.saliency(fr, out=out)
times = []
for _ in range(ITERS):
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.saliency(fr, out=out)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t)
times.sort()
print('this package, svc_saliency_u8 (LANCZOS + network + u8 quantisation), one pass alone: median %.3f ms (min %.3f) = %.0f frames/s'
      % (times[len(times) // 2] * 1e3, times[0] * 1e3, B / times[len(times) // 2]))
eng.close()
