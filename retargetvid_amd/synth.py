"""Synthetic inputs shaped like the reference's workloads.

No DHF1K video exists on the build or GPU boxes, so benchmark and test frames are
generated: 1-3 moving Gaussian blobs (sigma 20-60 px at 640x360) over low-amplitude
uniform noise.  Deterministic for a given seed (NumPy RandomState)."""
import numpy as np


def blob_frames(n, h=360, w=640, seed=0, n_blobs=None, dtype=np.uint8):
    """-> uint8 [n,h,w,3] RGB."""
    rng = np.random.RandomState(seed)
    nb = int(rng.randint(1, 4)) if n_blobs is None else n_blobs
    s = max(h, w) / 640.0
    cx = rng.uniform(0.15 * w, 0.85 * w, nb)
    cy = rng.uniform(0.2 * h, 0.8 * h, nb)
    vx = rng.uniform(-3, 3, nb) * s
    vy = rng.uniform(-2, 2, nb) * s
    sig = rng.uniform(20, 60, nb) * s
    amp = rng.uniform(150, 200, nb)
    col = rng.uniform(0.7, 1.0, (nb, 3))
    ys = np.arange(h, dtype=np.float32)[:, None]
    xs = np.arange(w, dtype=np.float32)[None, :]
    out = np.empty((n, h, w, 3), dtype)
    for i in range(n):
        img = rng.uniform(10, 40, (h, w, 3)).astype(np.float32)
        for b in range(nb):
            x0 = (cx[b] + vx[b] * i) % w
            y0 = (cy[b] + vy[b] * i) % h
            g = amp[b] * np.exp(-((xs - x0) ** 2 + (ys - y0) ** 2) / (2 * sig[b] ** 2))
            img += g[:, :, None] * col[b][None, None, :]
        out[i] = np.clip(img, 0, 255).astype(dtype)
    return out
