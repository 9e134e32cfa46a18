"""BASELINE.json configs[3] and configs[4] on one MI355X (GPU box helper).

  config 4: uint8[128,1080,1920,3] -> ingest down-scale -> saliency -> crop centres.  Reference semantics give
            the same 140x250 saliency frame and 256x416 network input as 640x360, so the conv work is 128 frames
            of the same network and the 1080p cost is the down-scale's read of the source frames.
  config 5: uint8[*,2160,3840,3] random stream, throughput only.  Two numbers: frames already in HBM
            (kernel-only) and frames fed from pinned host memory over PCIe (system number).

Prints one JSON object per measurement; ``--out FILE`` also writes them as a JSON list.
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # layer_work(): the algorithmic FLOPs of the pw class
from retargetvid_amd import ops, synth, smartVidCrop as S

HBM_PEAK_GBS = 8000.0


def timed(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


def class_ms(eng, fn, classes):
    out = {}
    for k in classes:
        eng.profile_enable(k)
        eng.profile_read()
        fn()
        torch.cuda.synchronize()
        out[k] = round(eng.profile_read()[0], 4)
    eng.profile_enable(None)
    return out


def touched_bytes(h, w, sh, sw, sector=32):
    """Bytes of ONE source frame the INTER_LINEAR down-scale actually reads: the four taps of every output pixel
    (exact), and the distinct 32-byte sectors they fall into (what the memory system moves at least)."""
    def taps(src, dst):
        f = (np.arange(dst) + 0.5) * (src / dst) - 0.5
        s0 = np.clip(np.floor(f).astype(np.int64), 0, src - 1)
        return s0, np.minimum(s0 + 1, src - 1)
    x0, x1 = taps(w, sw)
    y0, y1 = taps(h, sh)
    rows = np.unique(np.concatenate([y0, y1]))
    cols = np.unique(np.concatenate([x0, x1]))
    exact = len(rows) * len(cols) * 3
    sect = set()
    for c in cols:
        sect.add((3 * c) // sector)
        sect.add((3 * c + 2) // sector)
    # a row of w*3 bytes starts at an arbitrary sector phase; the count per row is the same up to one sector
    return exact, len(rows) * len(sect) * sector


def npts(eng, frames, CP):
    maps = eng.saliency(eng.resize_frames(frames, 140, 250))
    eng.threshold_(maps, CP['t_threshold'])
    return round(float((maps > 0).sum().item()) / maps.shape[0], 1)


def config4(batch, chunk, steps, warm):
    os.environ['SVC_CHUNK'] = str(chunk)
    eng = ops.Engine(seed=0)
    CP = S.sc_init_crop_params()
    frames = synth.LazyBlobVideo(batch, 1080, 1920, seed=4).select(range(batch))
    flags = np.zeros(batch, np.uint8)
    flags[:2] = 1

    def step():
        small = eng.resize_frames(frames, 140, 250)
        maps = eng.saliency(small)
        eng.threshold_(maps, CP['t_threshold'])
        return eng.cluster_center_(maps, flags, CP)

    ms = timed(step, steps, warm)
    per = class_ms(eng, step, ops.Engine.KERNEL_CLASSES)
    work = bench.layer_work(batch)
    src_bytes = frames.numel()
    res = dict(config='1080p frames, batch=%d, conv MFMA stress' % batch, chunk=chunk, ms_per_step=round(ms, 3),
               frames_per_s=round(batch / ms * 1e3, 1), class_ms_per_step=per,
               pw_tflops=round(work['pw_flops'] / (per['pw'] * 1e-3) / 1e12, 2),
               pw_frac_of_f32_mfma_peak=round(work['pw_flops'] / (per['pw'] * 1e-3) / 1e12 / bench.MFMA_F32_PEAK_TFLOPS, 4),
               mean_points_per_map=npts(eng, frames, CP),
               resize_source_GB=round(src_bytes / 1e9, 3),
               resize_touched_GB=round(batch * touched_bytes(1080, 1920, 140, 250)[1] / 1e9, 4),
               resize_GBs_touched=round(batch * touched_bytes(1080, 1920, 140, 250)[1] / (per['resize'] * 1e-3) / 1e9, 1),
               resize_frac_of_hbm_peak=round(batch * touched_bytes(1080, 1920, 140, 250)[1] / (per['resize'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               note='touched = the distinct 32-byte sectors holding the four taps of every output pixel (INTER_LINEAR at '
                    '7.68:1 reads 2 of every ~7.7 rows and 2 of every ~7.7 pixels) + the output; the source frames '
                    'themselves are %.3f GB' % (src_bytes / 1e9))
    eng.close()
    return res


def config5(batch, steps, warm):
    os.environ['SVC_CHUNK'] = '32'
    eng = ops.Engine(seed=0)
    CP = S.sc_init_crop_params()
    # moving blobs over noise like every other synthetic input: uniform random pixels through seeded random
    # weights give maps with tens of thousands of foreground pixels, the O(N^2) worst case of the cluster filter
    dev = synth.LazyBlobVideo(batch, 2160, 3840, seed=5).select(range(batch))
    host = torch.empty((batch, 2160, 3840, 3), dtype=torch.uint8).pin_memory()
    host.copy_(dev)
    stage = torch.empty_like(dev)
    flags = np.zeros(batch, np.uint8)

    def tail(frames):
        small = eng.resize_frames(frames, 140, 250)
        maps = eng.saliency(small)
        eng.threshold_(maps, CP['t_threshold'])
        return eng.cluster_center_(maps, flags, CP)

    ms_dev = timed(lambda: tail(dev), steps, warm)

    def fed():
        stage.copy_(host, non_blocking=True)
        return tail(stage)

    ms_host = timed(fed, steps, warm)

    # the product's host-fed ingest (smartVidCrop._HostFeed: pinned double buffers, H2D on a side stream, selection
    # before the copy, down-scale per chunk on the compute stream) followed by the rest of the chain
    feed = S._HostFeed(eng)
    host_np = host.numpy()
    idx = list(range(batch))

    def fed_product(src):
        small = feed.downscale(src, idx, 140, 250)
        maps = eng.saliency(small)
        eng.threshold_(maps, CP['t_threshold'])
        return eng.cluster_center_(maps, flags, CP)

    ms_ovl = timed(lambda: fed_product(host_np), steps, warm)      # pageable-style source: gathered into the pinned slots
    ms_pin = timed(lambda: fed_product(host), steps, warm)         # pinned source: copied from where it lies
    ms_copy = timed(lambda: stage.copy_(host, non_blocking=True), steps, warm)
    per = class_ms(eng, lambda: tail(dev), ('resize', 'pw'))
    nbytes = dev.numel()
    res = dict(config='Synthetic random 4K frame stream, throughput only, 1 GPU of the 8', batch=batch,
               frames_in_hbm=dict(ms_per_step=round(ms_dev, 3), frames_per_s=round(batch / ms_dev * 1e3, 1)),
               frames_from_pinned_host=dict(ms_per_step=round(ms_host, 3), frames_per_s=round(batch / ms_host * 1e3, 1),
                                            h2d_ms=round(ms_copy, 3), h2d_GBs=round(nbytes / (ms_copy * 1e-3) / 1e9, 1)),
               frames_from_host_product_ingest=dict(ms_per_step=round(ms_ovl, 3), frames_per_s=round(batch / ms_ovl * 1e3, 1)),
               frames_from_pinned_host_product_ingest=dict(ms_per_step=round(ms_pin, 3), frames_per_s=round(batch / ms_pin * 1e3, 1)),
               resize_ms=per['resize'],
               resize_GBs_touched=round(batch * touched_bytes(2160, 3840, 140, 250)[1] / (per['resize'] * 1e-3) / 1e9, 1),
               mean_points_per_map=npts(eng, dev, CP),
               note='fed from the host the stream is bound by PCIe (24.9 MB per frame); product ingest = pinned double '
                    'buffers of 96 MB (3 frames of 4K each), copies on a side stream')
    eng.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    out = []
    for chunk in (32, 128):
        out.append(config4(128, chunk, a.steps, a.warmup))
        print(json.dumps(out[-1]), flush=True)
    out.append(config5(32, a.steps, a.warmup))
    print(json.dumps(out[-1]), flush=True)
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
