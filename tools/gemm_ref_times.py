"""How fast does the vendor SGEMM (torch.mm, fp32) run the pointwise-conv shapes?  Reference point only."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(3328, 960, 160), (3328, 160, 960), (3328, 320, 1280), (3328, 1296, 256), (13312, 384, 768), (13312, 768, 128),
          (13312, 576, 160), (53248, 192, 384), (53248, 384, 64), (851968, 16, 96), (212992, 144, 24), (53248, 192, 32)]
for M, K, N in shapes:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda')
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): c = a @ b.t()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print('M=%7d K=%5d N=%5d  %8.1f us  %6.1f TFLOP/s  %6.2f TB/s' % (M, K, N, dt * 1e6, 2.0 * M * K * N / dt / 1e12, 4.0 * (M * K + M * N + K * N) / dt / 1e12))
