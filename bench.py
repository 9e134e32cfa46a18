#!/usr/bin/env python3
"""Headline benchmark: end-to-end saliency+crop frames/s on synthetic 640x360 frame batches.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step is one pass of the hot path over one batch of B=32 device-resident 640x360 RGB
frames (BASELINE.json configs[1]): ingest down-scale -> UNISAL static saliency ->
threshold -> cluster filter with cut blend -> centre of focus (all HIP, through the C ABI),
then one D2H of the 32 centres and the crop-box arithmetic on the host.  Frames shard
across ranks with no data-path collective ("weak" scaling: 32 frames per GPU per step); the
only exchange is the all_gather of the final boxes after the timed region.

Rank 0 prints ONE JSON line.  `roofline` describes the kernel class that takes the most
time inside the timed region (HIP events recorded by the library around each launch of that
class on its stream); `cpu_baseline` is the oracle (CPU restatement of the same path) timed
on this node's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

# one HIP stream per in-flight batch: give the runtime enough hardware queues that the streams do not
# share one (ROCclr default is 4; with it the same run is ~20 % slower) — must be set before HIP starts
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from retargetvid_amd import dist as svc_dist, ops, smartVidCrop as S, synth, weights   # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA


def layer_work(batch, nh=256, nw=416, h=140, w=250):
    """Algorithmic FLOPs / bytes per step for the kernel classes.  Spatial sizes of
    the static SALICON graph for a 16:9 input."""
    sp = {1: (nh // 2, nw // 2), 2: (nh // 4, nw // 4), 3: (nh // 8, nw // 8), 4: (nh // 16, nw // 16), 5: (nh // 32, nw // 32)}
    pw = []      # (pixels, cin, cout)
    dw = []      # (in pixels, out pixels, channels)
    level, inp = 1, 32
    stages = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
    idx = 1
    for t, c, n, s in stages:
        for i in range(n):
            stride = s if i == 0 else 1
            hh, ww = sp[level]
            px = hh * ww
            if t != 1:
                pw.append((px, inp, inp * t))
            tap = idx in (7, 14)
            out_level = level + 1 if stride == 2 else level
            opx = px if (stride == 1 or tap) else sp[out_level][0] * sp[out_level][1]
            dw.append((px, opx, inp * t))
            pw.append((opx, inp * t, c))
            level, inp, idx = out_level, c, idx + 1
    p5, p4, p3 = sp[5][0] * sp[5][1], sp[4][0] * sp[4][1], sp[3][0] * sp[3][1]
    pw += [(p5, 320, 1280), (p4, 160, 320), (p4, 320, 128), (p3, 64, 128), (p3, 128, 64), (p5, 1296, 256),
           (p4, 384, 768), (p4, 768, 128), (p3, 192, 384), (p3, 384, 64)]
    dw += [(p5, p5, 1296), (p4, p4, 768), (p3, p3, 384)]
    pw_flops = 2.0 * sum(m * k * n for m, k, n in pw) * batch
    pw_bytes = 4.0 * sum(batch * m * (k + n) + k * n for m, k, n in pw)
    dw_bytes = 4.0 * sum(batch * (a + b) * c + 10 * c for a, b, c in dw)
    return dict(pw_flops=pw_flops, pw_bytes=pw_bytes, dw_bytes=dw_bytes, pw_launches=len(pw), dw_launches=len(dw),
                map_bytes=2.0 * batch * h * w)


def host_boxes(xy, ratio='1:3'):
    """Centres (saliency-map pixels) -> crop boxes for a 640x360 video; NaN centres take the previous one."""
    VD = dict(w_orig=640, h_orig=360, h_process=140, w_process=250, fc=len(xy))
    S.sc_calc_dest_size(VD, {'out_ratio': ratio})
    xs, ys, last = [], [], (125.0, 70.0)
    for x, y in xy:
        if x == x:
            last = (float(x), float(y))
        xs.append(last[0])
        ys.append(last[1])
    VD['dxs'], VD['dys'] = xs, ys
    return S.sc_compute_bb(VD, {})['bbs']


def cpu_baseline(sd, frames_u8, CP, flags):
    """The oracle on this node's host cores over a bounded sample of the same workload."""
    from oracle import cv_ref, tail_ref as T, unisal_ref as U
    n = frames_u8.shape[0]
    t0 = time.perf_counter()
    small = np.stack([cv_ref.resize_linear_u8(f, 140, 250) for f in frames_u8])
    maps = U.saliency_u8(sd, small)
    T.threshold(maps, CP['t_threshold'])
    for i in range(n):
        maps[:, :, i] = T.clustering_filt(maps[:, :, i], CP)
        if i + 1 < n and flags[i]:
            maps[:, :, i + 1] = T.blend_next(maps[:, :, i], maps[:, :, i + 1])
    dx, dy = T.centers(maps, CP)
    host_boxes([(np.nan, np.nan) if x is None else (x, y) for x, y in zip(dx, dy)])
    dt = time.perf_counter() - t0
    return n / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--pipeline', type=int, default=4, help='batches in flight per GPU (engines / HIP streams)')
    ap.add_argument('--cpu-sample', type=int, default=160, help='frames of the CPU baseline sample (0 = skip)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, barrier, MAX all-reduce, box gather) with any
    # world size, including 1 under torch.distributed.run -- a way to exercise it on a single-GPU box
    dist_on = world > 1 or os.environ.get('BENCH_FORCE_DIST', '0') == '1'
    if dist_on:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    B = args.batch
    P = max(1, args.pipeline)
    sd = weights.make_synthetic_state_dict(0)
    CP = S.sc_init_crop_params()
    CP['out_ratio'] = '1:3'
    frames_host = synth.blob_frames(B, 360, 640, seed=100 + rank)
    frames = torch.from_numpy(frames_host).to(dev)
    flags = np.zeros(B, np.uint8)
    if os.environ.get('BENCH_NO_BLEND', '0') != '1':   # (diagnostic: a batch without a cut has one tail round instead of three)
        flags[:2] = 1                  # the batch starts a shot: maps 0,1 blend into 1,2 (smartVidCrop.py:2369-2373)

    class Slot:
        """One in-flight step: its own engine (weights + workspace), HIP stream and pinned result buffer,
        so the low-occupancy clustering tail of one batch overlaps the network of the next."""
        def __init__(self):
            self.eng = ops.Engine(sd)
            self.stream = torch.cuda.Stream(device=dev)
            self.xy_host = torch.empty((B, 2), dtype=torch.float64).pin_memory()
            self.done = torch.cuda.Event()
            self.pending = False

        def enqueue(self):
            with torch.cuda.stream(self.stream):
                small = self.eng.resize_frames(frames, 140, 250)
                maps = self.eng.saliency(small)
                self.eng.threshold_(maps, CP['t_threshold'])
                xy = self.eng.cluster_center_(maps, flags, CP)
                self.xy_host.copy_(xy, non_blocking=True)
                self.done.record(self.stream)
            self.pending = True

        def finish(self):
            self.done.synchronize()
            self.pending = False
            return host_boxes(self.xy_host.numpy())

    slots = [Slot() for _ in range(P)]
    torch.cuda.synchronize()

    def run(steps):
        boxes = None
        for s in range(steps):
            sl = slots[s % P]
            if sl.pending:
                boxes = sl.finish()
            sl.enqueue()
        for sl in slots:
            if sl.pending:
                boxes = sl.finish()
        return boxes

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    boxes = run(max(args.warmup, P))
    # find the kernel class that takes the most device time (one profiled, un-pipelined, untimed step per class)
    eng = slots[0].eng
    per_class = {}
    for k in eng.KERNEL_CLASSES:
        eng.profile_enable(k)
        slots[0].enqueue()
        slots[0].finish()
        per_class[k] = eng.profile_read()
    dominant = max(per_class, key=lambda k: per_class[k][0])
    live = os.environ.get('BENCH_LIVE_PROFILE', '1') != '0'      # diagnostic: cost of the in-library events
    for sl in slots:
        sl.eng.profile_enable(dominant if live else None)
        sl.eng.profile_read()

    barrier()
    t0 = time.perf_counter()
    boxes = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    dom_ms, dom_launches = 0.0, 0
    for sl in slots:
        ms, cnt = sl.eng.profile_read()
        dom_ms += ms
        dom_launches += cnt
        sl.eng.profile_enable(None)
    if not live:                                                  # fall back to the isolated per-class measurement
        dom_ms, dom_launches = per_class[dominant][0] * args.steps, per_class[dominant][1] * args.steps

    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        work = layer_work(B)
        steps = max(args.steps, 1)
        ms_per_launch = dom_ms / max(dom_launches, 1)
        if dominant == 'pw':
            tfl = work['pw_flops'] * steps / (dom_ms * 1e-3) / 1e12
            gbs = work['pw_bytes'] * steps / (dom_ms * 1e-3) / 1e9
            roof = dict(bound='mfma', achieved=round(tfl, 3), peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                        frac=round(tfl / MFMA_F32_PEAK_TFLOPS, 5), traffic=None,
                        hbm_achieved_GBs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 5))
        else:
            byt = {'dw': work['dw_bytes'], 'lanczos': B * (140 * 250 * 3 + 256 * 416 * 12.0),
                   'stem': B * (256 * 416 * 12.0 + 128 * 208 * 128.0), 'resize': B * (360 * 640 * 3 + 140 * 250 * 3.0),
                   'smooth': B * (32 * 52 * 4 + 140 * 250 * 9.0)}.get(dominant, work['map_bytes'])
            gbs = byt * steps / (dom_ms * 1e-3) / 1e9
            roof = dict(bound='hbm', achieved=round(gbs, 3), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(gbs / HBM_PEAK_GBS, 6), traffic=None)
        # un-pipelined view of the same class (the profiling pass before the timed region) and the PMC traffic
        iso_ms, iso_n = per_class[dominant]
        if dominant == 'pw' and iso_ms > 0:
            roof.update(achieved_isolated=round(work['pw_flops'] / (iso_ms * 1e-3) / 1e12, 3),
                        frac_isolated=round(work['pw_flops'] / (iso_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 5))
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')) as fp:
                roof['traffic'] = json.load(fp)['classes'][dominant]['hbm_bytes_per_launch']
        except Exception:
            pass
        roof.update(kernel=dominant, launches=dom_launches, avg_launch_ms=round(ms_per_launch, 4),
                    class_ms_per_step={k: round(v[0], 3) for k, v in per_class.items()})
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            torch.set_num_threads(min(16, os.cpu_count() or 1))    # batch-1 convs stop scaling (and collapse) beyond this
            nb = max(1, args.cpu_sample // B)
            secs, nfr = 0.0, 0
            for b in range(nb):                                    # whole steps of the same workload, new frames each
                fh = frames_host if b == 0 else synth.blob_frames(B, 360, 640, seed=200 + b)
                fps_b, s_b = cpu_baseline(sd, fh, CP, flags)
                secs += s_b
                nfr += B
            cpu = dict(value=round(nfr / secs, 3), unit='frames/s', cores=torch.get_num_threads(), kind='port',
                       sample='%d steps of the same workload (%d frames), oracle/ (PyTorch-CPU fp32 forward at batch 1, '
                              'NumPy tail), %.1f s' % (nb, nfr, secs))
        value = world * B * args.steps / dt
        out = dict(metric='frames/sec end-to-end saliency+crop on 640x360', value=round(value, 2), unit='frames/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / max(args.steps, 1) * 1e3, 4),
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='Single 640x360 video, batch=32 frames, UNISAL saliency + crop on 1 MI355X',
                               batch_per_gpu=B, frame='640x360x3 u8', saliency_map='140x250 u8', network_input='256x416',
                               weights='synthetic seed 0 (weights.make_synthetic_state_dict)',
                               video_frames_per_s=round(value * CP['skip'], 1), batches_in_flight=P,
                               parallelism='frames sharded, dp%d' % world),
                   roofline=roof, cpu_baseline=cpu)
        print(json.dumps(out), flush=True)
    if dist_on:
        # the path's one exchange: every rank ends up with all crop windows (outside the timed region)
        counts = [B] * world
        allb = svc_dist.gather_boxes({i: np.asarray(boxes, np.int32) for i in svc_dist.shard_videos(counts, world)[rank]}, counts)
        assert len(allb) == world and all(v.shape == (B, 4) for v in allb.values())
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    for sl in slots:
        sl.eng.close()


if __name__ == '__main__':
    main()
