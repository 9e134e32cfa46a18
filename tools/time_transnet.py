"""Throughput of the device TransNet V1 (GPU box helper): frames/s over a 2 000-frame video, and the oracle on the host."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import transnetv1_handler as Hd, weights
from oracle import transnet_ref as R
sd = weights.make_transnet_state_dict(0)
net = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=sd, windows_per_call=int(os.environ.get('WPC', '8')))
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (2000, 27, 48, 3)).astype(np.uint8)).cuda()
net.predict_video(fr)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): p = net.predict_video(fr)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print('device: 2000 frames (40 windows of 100) in %.1f ms = %.0f frames/s; %.1f TFLOP/s of the 110 GFLOP per window' % (dt * 1e3, 2000 / dt, 40 * 110e9 / dt / 1e12))
if os.environ.get('CPU', '1') == '1':
    torch.set_num_threads(16)
    sub = fr[:200].cpu().numpy()
    t = time.perf_counter(); R.predict_video(sd, sub); dt = time.perf_counter() - t
    print('oracle on the host (16 threads): 200 frames in %.2f s = %.0f frames/s' % (dt, 200 / dt))
