"""Synthetic inputs shaped like the reference's workloads.

No DHF1K video exists on the build or GPU boxes, so benchmark and test frames are
generated: 1-3 moving Gaussian blobs (sigma 20-60 px at 640x360) over low-amplitude
uniform noise.  Deterministic for a given seed (NumPy RandomState)."""
import numpy as np


def blob_frames(n, h=360, w=640, seed=0, n_blobs=None, dtype=np.uint8, sigma=(20, 60)):
    """-> uint8 [n,h,w,3] RGB.  sigma: range of the blobs' standard deviation in pixels at 640x360 (the benchmark uses
    larger blobs than the tests so that a thresholded map holds the 1-6 k points SURVEY.md 8(d) prescribes)."""
    rng = np.random.RandomState(seed)
    nb = int(rng.randint(1, 4)) if n_blobs is None else n_blobs
    s = max(h, w) / 640.0
    cx = rng.uniform(0.15 * w, 0.85 * w, nb)
    cy = rng.uniform(0.2 * h, 0.8 * h, nb)
    vx = rng.uniform(-3, 3, nb) * s
    vy = rng.uniform(-2, 2, nb) * s
    sig = rng.uniform(sigma[0], sigma[1], nb) * s
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


def blob_tracks(n, h=360, w=640, seed=0, n_blobs=None, sigma=(20, 60)):
    """Where blob_frames(n, h, w, seed, ...) puts its blobs: -> (x [n, nb], y [n, nb], sigma [nb], amplitude [nb]) in pixels
    of the h x w frame (the same RandomState draws, without rendering)."""
    rng = np.random.RandomState(seed)
    nb = int(rng.randint(1, 4)) if n_blobs is None else n_blobs
    s = max(h, w) / 640.0
    cx = rng.uniform(0.15 * w, 0.85 * w, nb)
    cy = rng.uniform(0.2 * h, 0.8 * h, nb)
    vx = rng.uniform(-3, 3, nb) * s
    vy = rng.uniform(-2, 2, nb) * s
    sig = rng.uniform(sigma[0], sigma[1], nb) * s
    amp = rng.uniform(150, 200, nb)
    i = np.arange(n)[:, None]
    return (cx[None] + vx[None] * i) % w, (cy[None] + vy[None] * i) % h, sig, amp


class LazyBlobVideo:
    """A synthetic video whose frames are generated on demand on the GPU (torch), so a
    RetargetVid-sized run (122 684 frames of 640x360) never materialises 85 GB of pixels.
    ``len(v)`` = frame count, ``v.select(idx)`` -> uint8 CUDA tensor [len(idx),h,w,3].
    Deterministic for a given seed and device type."""

    def __init__(self, n, h=360, w=640, seed=0, device='cuda'):
        import torch
        self.n, self.h, self.w, self.seed, self.device = n, h, w, seed, torch.device(device)
        rng = np.random.RandomState(seed)
        nb = int(rng.randint(1, 4))
        s = max(h, w) / 640.0
        self.p = dict(cx=rng.uniform(0.15 * w, 0.85 * w, nb), cy=rng.uniform(0.2 * h, 0.8 * h, nb),
                      vx=rng.uniform(-3, 3, nb) * s, vy=rng.uniform(-2, 2, nb) * s,
                      sig=rng.uniform(20, 60, nb) * s, amp=rng.uniform(150, 200, nb),
                      col=rng.uniform(0.7, 1.0, (nb, 3)))

    def __len__(self):
        return self.n

    accepts_device_index = True        # select(idx, index=<CUDA int64 tensor>): smartVidCrop._small_frames passes its own

    def select(self, idx, index=None):
        """index: the frame numbers as a CUDA tensor, if the caller has them there already (no host-device copy here)."""
        import torch
        dev = self.device
        idx = list(idx)
        if index is not None:
            t = index.to(torch.float32).view(-1, 1, 1)
        elif len(idx) > 1 and all(idx[i + 1] - idx[i] == idx[1] - idx[0] for i in range(len(idx) - 1)) and idx[1] > idx[0]:
            t = torch.arange(idx[0], idx[0] + (idx[1] - idx[0]) * len(idx), idx[1] - idx[0], dtype=torch.float32, device=dev).view(-1, 1, 1)
        else:
            t = torch.as_tensor(idx, dtype=torch.float32, device=dev).view(-1, 1, 1)       # (synchronises: pageable copy)
        ys = torch.arange(self.h, dtype=torch.float32, device=dev).view(1, -1, 1)
        xs = torch.arange(self.w, dtype=torch.float32, device=dev).view(1, 1, -1)
        # deterministic low-amplitude texture (a hash of position and frame) instead of a host RNG stream
        tex = torch.frac(torch.sin(xs * 12.9898 + ys * 78.233 + t * 37.719 + self.seed) * 43758.5453) * 30.0 + 10.0
        img = tex.unsqueeze(-1).expand(-1, -1, -1, 3).clone()
        p = self.p
        for b in range(len(p['cx'])):
            x0 = torch.remainder(p['cx'][b] + p['vx'][b] * t, self.w)
            y0 = torch.remainder(p['cy'][b] + p['vy'][b] * t, self.h)
            g = p['amp'][b] * torch.exp(-((xs - x0) ** 2 + (ys - y0) ** 2) / (2 * p['sig'][b] ** 2))
            for c in range(3):                                       # (scalars: no host-device copy of the colour)
                img[..., c].add_(g, alpha=float(p['col'][b][c]))
        return img.clamp_(0, 255).to(torch.uint8).contiguous()


class ResidentBlobVideo:
    """The frames of a LazyBlobVideo that a run will ask for, generated ONCE and kept in HBM (uint8 [k,h,w,3]); select(idx)
    gathers them.  bench.py's convention -- inputs resident in HBM when the timed region starts -- for a whole job: the
    200-video set needs 20 964 frames = 14.5 GB, where generating them on the fly costs the GPU ~40 element-wise passes
    over every frame inside the run.  Asking for a frame that was not generated raises."""

    accepts_device_index = False

    def __init__(self, n, indices, h=360, w=640, seed=0, device='cuda', piece=64):
        import torch
        self.n, self.h, self.w = n, h, w
        src = LazyBlobVideo(n, h, w, seed=seed, device=device)
        idx = [int(i) for i in indices]
        self.row = {f: r for r, f in enumerate(idx)}
        self.frames = torch.cat([src.select(idx[s:s + piece]) for s in range(0, len(idx), piece)]) if idx else \
            torch.empty((0, h, w, 3), dtype=torch.uint8, device=device)

    def __len__(self):
        return self.n

    def select(self, idx, index=None):
        rows = [self.row[int(i)] for i in idx]
        if rows and rows == list(range(rows[0], rows[0] + len(rows))):
            return self.frames[rows[0]:rows[0] + len(rows)]               # a run of consecutive rows: a view, no gather
        import torch
        return self.frames[torch.as_tensor(rows, dtype=torch.int64).pin_memory().to(self.frames.device, non_blocking=True)]


class HostSelectedVideo:
    """The frames a run will select, in PINNED HOST memory (uint8 [k,h,w,3]): the "decoded frames live on the host" case of
    BASELINE config 3 without a decoder -- every selected frame crosses PCIe inside the job, through smartVidCrop._HostFeed
    (pinned source: copied from where it lies, then down-scaled on the device).  Built from a ResidentBlobVideo (one
    device-to-host copy, outside any timed region).  ``rows(idx)``: positions of the frame numbers idx in ``pinned``."""

    def __init__(self, resident, pinned=None):
        """pinned: a pinned uint8 tensor of the frames' shape to fill (a slice of one big allocation: page-locking 14.5 GB in 200
        pieces takes 10 s, in one piece a few)."""
        import torch
        self.n, self.h, self.w = resident.n, resident.h, resident.w
        self.row = resident.row
        self.pinned = pinned if pinned is not None else torch.empty(tuple(resident.frames.shape), dtype=torch.uint8).pin_memory()
        assert self.pinned.is_pinned() and tuple(self.pinned.shape) == tuple(resident.frames.shape)
        self.pinned.copy_(resident.frames)

    def __len__(self):
        return self.n

    def rows(self, idx):
        return [self.row[int(i)] for i in idx]


def retargetvid_cuts(vid, n):
    """The synthetic shot starts of video `vid` (n frames) in the RetargetVid-shaped job of bench.py's config 3,
    tools/run_config3.py and the tests: 0-3 cuts at seeded positions.  -> trans_inds (smartVidCrop.py:560-573)."""
    rng = np.random.RandomState(vid)
    return sorted(set([0] + [int(c) for c in rng.randint(20, max(21, n - 20), rng.randint(0, 4))])) + [n]


def windows_crc32(boxes_by_ratio, ratios, n_videos):
    """One checksum over the crop windows of a job: int32 boxes of every video, ratio-major."""
    import zlib
    c = 0
    for r in ratios:
        for i in range(n_videos):
            c = zlib.crc32(np.ascontiguousarray(boxes_by_ratio[r][i], np.int32).tobytes(), c)
    return c & 0xffffffff
