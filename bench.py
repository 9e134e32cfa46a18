#!/usr/bin/env python3
"""Headline benchmark: end-to-end saliency+crop frames/s on synthetic 640x360 frame batches.

  python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ...` on this same file as a CHILD (before anything here has touched the
GPU; nothing is exec'ed) and exits with its code.  Under torch.distributed.run (the driver's form) every rank runs
main() directly; --gpus must then equal WORLD_SIZE.

A step is one pass of the hot path over one batch of B=32 device-resident 640x360 RGB
frames (BASELINE.json configs[1]): ingest down-scale -> UNISAL static saliency ->
threshold -> cluster filter with cut blend -> centre of focus (all HIP, through the C ABI),
then one D2H of the 32 centres and the crop-box arithmetic on the host.  Frames shard
across ranks with no data-path collective ("weak" scaling: 32 frames per GPU per step); the
only exchange is the all_gather of the final boxes after the timed region.

Every batch starts a shot: its maps 0, 1, 2 form a blend chain (each is clustered after its predecessor has been), the
other 29 maps are independent.  The timed loop drives the product's streaming scheduler (retargetvid_amd/pipeline.py:
StreamPipeline, one per HIP stream): a call processes the maps whose predecessor is final and leaves the rest of a chain
to the stream's next calls (SVC_MAP_HELD, include/svc.h), so a call has one tail round instead of three serial ones;
maps and centres are those of the reference's loop (tests/test_gpu_pipeline.py::test_stream_pipeline_equals_the_oracle_loop).
Every batch is completed inside the timed region (finish() runs the carried maps out).  BENCH_CARRY=0 selects the plain
call (the chains run out in rounds inside every call); config.one_batch_in_flight is that plain call with one batch in
flight.  The synthetic frames are sized so that a thresholded map holds ~2 k points (1-6 k);
config.points_per_map reports what the run saw.  The K-step timed region is repeated --repeats times; ms_per_step / value
are the median region, config.repeats lists all of them.

Rank 0 prints ONE JSON line.  `roofline` describes the kernel class that takes the most device time
(HIP events recorded by the library around each launch of that class, on the stream it is launched on):
  achieved / frac          algorithmic FLOPs (or bytes) of the class per step / the summed durations of its launches
                           in an UN-PIPELINED step (one batch in flight: launch durations that do not overlap anything,
                           what `rocprofv3 --kernel-trace --stats` of `bench.py --pipeline 1` shows per kernel)
  *_in_flight              (BENCH_LIVE_PROFILE=1 only) the same quotient over the events of the timed region, where several batches are in flight
                           on their own streams: launch durations then include time shared with other streams' kernels
                           and their sum exceeds the wall clock; kept for reference, never the headline fraction
  frac_wall                algorithmic work of the class per step / ms_per_step: what the class achieves per wall-clock
                           second of the pipelined run (a lower bound: every other kernel shares that time)
  traffic                  HBM bytes per launch of the class from the stored rocprofv3 PMC passes (file named in
                           traffic_source; not re-measured by this run)
`cpu_baseline` is the oracle (CPU restatement of the same path) timed on this node's host cores on a bounded
sample of the same workload.
`cpu_baseline.network_under_pytorch_rocm` (BENCH_TORCH_BASELINE=0 skips it): the NETWORK alone the way the reference runs it -- eager
PyTorch modules, here under PyTorch-ROCm (MIOpen / rocBLAS) -- on this GPU, forward only: a second reported baseline, same rules.
`config.config3` (BENCH_CONFIG3=0 skips it) is BASELINE.json's configs[2] run ONCE MORE OUTSIDE the timed region through the
product's job code: the 200-video RetargetVid-shaped synthetic set (real frame counts, 1:3 and 3:1), videos sharded over the
ranks (dist.crop_job), every rank's share through the job-level scheduler (retargetvid_amd/scheduler.py: frames of
consecutive videos packed into full 32-frame chunks), boxes all_gathered, scored by the evaluator counterpart.  The frames
the job selects are generated before its clock starts and lie in HBM, like this benchmark's own batch.
"""
import argparse
import json
import os
import sys
import time

# one HIP stream per in-flight batch: give the runtime enough hardware queues that the streams do not
# share one (ROCclr default is 4; with it the same run is ~20 % slower) — must be set before HIP starts
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')      # (16 streams run at once with 16; with 4 the default layout drops to 2.3 ms per step)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--pipeline', type=int, default=4, help='batches in flight per GPU (engines / HIP streams)')
    ap.add_argument('--cpu-sample', type=int, default=160, help='frames of the CPU baseline sample (0 = skip)')
    ap.add_argument('--iso-steps', type=int, default=3, help='un-pipelined profiling steps per kernel class')
    ap.add_argument('--repeats', type=int, default=15, help='timed regions of --steps steps each; the median is reported')
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as a child job and return its exit code.
    Runs before torch / HIP are imported in this process, and starts a child instead of replacing this process."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


if __name__ == '__main__' and 'WORLD_SIZE' not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        sys.exit(self_launch(_a))

import numpy as np      # noqa: E402
import torch            # noqa: E402

from retargetvid_amd import dist as svc_dist, ops, pipeline, smartVidCrop as S, synth, weights   # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF)
# the benchmark's frames: two blobs per frame, sized so that a map thresholded at 120 holds ~2 k points
BENCH_BLOBS = dict(n_blobs=2, sigma=(float(os.environ.get('BENCH_SIGMA_LO', 30)), float(os.environ.get('BENCH_SIGMA_HI', 44))))
if os.environ.get('BENCH_R02_WORKLOAD', '0') == '1':      # the frames of rounds 1-2 (1-3 blobs, sigma 20-60: ~650 points per map), for like-for-like comparisons
    BENCH_BLOBS = {}


def layer_work(batch, nh=256, nw=416, h=140, w=250, front_fused=False):
    """Algorithmic FLOPs / bytes per step for the kernel classes.  Spatial sizes of
    the static SALICON graph for a 16:9 input.  front_fused: features.1 runs inside k_front, which
    is timed under the 'stem' class, so its depthwise and project leave the 'dw' / 'pw' work."""
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
            if not (front_fused and idx == 1):
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


def reference_formulation_on_gpu(sd, dev, batch, iters=10):
    """Part of the baseline leg: the NETWORK the way the reference runs it -- eager PyTorch modules (train.py:778-859, model.py:411-506;
    cuDNN on its hardware, MIOpen / rocBLAS under PyTorch-ROCm) -- on THIS GPU: oracle/unisal_ref.forward_logits (F.conv2d / F.batch_norm /
    F.interpolate, BatchNorm not folded, NCHW fp32) on `batch` frames of 256 x 416, forward only (no LANCZOS, no quantisation, no tail:
    those run on the CPU in the reference).  A reported baseline beside cpu_baseline, never the thing shipped; tools/torch_rocm_baseline.py
    is the stand-alone form."""
    from oracle import unisal_ref as U
    sd_dev = {k: (v if torch.is_tensor(v) else torch.as_tensor(v)).to(dev) for k, v in sd.items()}
    orig = U.gaussian_maps
    U.gaussian_maps = lambda g, h, w, scaling=6.0: orig(g.cpu(), h, w, scaling).to(dev)
    try:
        x = torch.randn(batch, 3, 256, 416, device=dev)
        t = time.perf_counter()
        U.forward_logits(sd_dev, x, (140, 250))
        torch.cuda.synchronize()
        first = time.perf_counter() - t
        for _ in range(2):
            U.forward_logits(sd_dev, x, (140, 250))
        ts = []
        for _ in range(iters):
            torch.cuda.synchronize()
            t = time.perf_counter()
            U.forward_logits(sd_dev, x, (140, 250))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        ts.sort()
        med = ts[len(ts) // 2]
        return dict(value=round(batch / med, 1), unit='frames/s', ms_per_pass=round(med * 1e3, 3), first_pass_s=round(first, 2),
                    what='the reference\'s formulation of the network (eager PyTorch-ROCm: MIOpen / rocBLAS, fp32 NCHW, BatchNorm not folded), '
                         'forward only, batch %d x 256 x 416, one pass alone on this GPU; compare with config.one_batch_in_flight and '
                         'batch_phase_ms_in_the_pipeline.network (which include LANCZOS and the u8 quantisation)' % batch,
                    torch=torch.__version__)
    finally:
        U.gaussian_maps = orig


def config3_job(world, rank, dist_on, dev, sd, lanes, rccl_init_s, mode='resident', shot_mx=None):
    """BASELINE configs[2] through the product's job code, timed between barriers (slowest rank).  -> dict for rank 0.
    mode 'resident': the frames the job selects are in HBM when the clock starts (bench.py's convention for `value`);
    'host_fed': they are in PINNED HOST memory and cross PCIe inside the job (smartVidCrop._HostFeed) -- the system number;
    'shot_net': the reference's VIDEO path (smartVidCrop.py:366-372): no trans_inds, every frame of every video resident in HBM
    (85 GB), TransNet V1 on the device inside the job decides the shots and with them the selection; shot_mx: the matrix
    pipe of its cells for this job (SVC_SHOT_MX; None = the library's default)."""
    from retargetvid_amd import evaluate as E, scheduler
    folder = os.path.join(ROOT, 'tests', 'golden', 'retargetvid')
    fcs = E.frame_counts(folder)
    vids = list(E.VID_INDS)
    counts = [fcs[v] for v in vids]
    CP = S.sc_init_crop_params()
    ratios = ('1:3', '3:1')

    def cuts_of(i):
        return synth.retargetvid_cuts(vids[i], counts[i])[:-1]

    t0 = time.perf_counter()
    resident, n_sel_mine = {}, 0
    shot_net = None
    for i in svc_dist.shard_videos(counts, world)[rank]:          # this rank's videos: the frames the job will select, resident in HBM
        if mode == 'shot_net':                                    # ... or every frame of the video (the selection is TransNet's to make)
            src = synth.LazyBlobVideo(counts[i], 360, 640, seed=vids[i], device=dev)
            resident[i] = torch.cat([src.select(range(a, min(counts[i], a + 64))) for a in range(0, counts[i], 64)])
            continue
        sel = S._select_frames(counts[i], counts[i], cuts_of(i) + [counts[i]], CP['skip'], CP['read_batch'])[0]
        resident[i] = synth.ResidentBlobVideo(counts[i], sel, seed=vids[i], device=dev)
        n_sel_mine += len(sel)
    if mode == 'host_fed':                                            # one page-locked allocation, a slice per video
        slab = torch.empty((n_sel_mine, 360, 640, 3), dtype=torch.uint8, pin_memory=True)
        at = 0
        for i in sorted(resident):
            k_ = int(resident[i].frames.shape[0])
            resident[i] = synth.HostSelectedVideo(resident[i], pinned=slab[at:at + k_])
            at += k_
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    if mode == 'shot_net':
        from retargetvid_amd import transnetv1_handler as TN
        # synthetic TransNet weights with the last layer biased to "no transition": random weights report arbitrary cuts --
        # including segmentations no video has (a scene before frame 0 ends, zero-length shots) that the host stages reject --
        # so every video comes out as ONE shot; the network's compute is that of a real checkpoint
        tsd = weights.make_transnet_state_dict(0)
        tsd['TransNet/dense_1/kernel'] = tsd['TransNet/dense_1/kernel'] * np.float32(0.02)
        tsd['TransNet/dense_1/bias'] = np.array([12.0, -12.0], np.float32)
        old_mx = os.environ.get('SVC_SHOT_MX')
        if shot_mx is not None:
            os.environ['SVC_SHOT_MX'] = shot_mx                      # read when the network's handle is created
        try:
            shot_net = TN.ShotTransNet(TN.ShotTransNetParams(), weights=tsd)
        finally:
            if shot_mx is not None:
                if old_mx is None:
                    os.environ.pop('SVC_SHOT_MX', None)
                else:
                    os.environ['SVC_SHOT_MX'] = old_mx
        shot_pipe = shot_net.matrix_pipe()

    def make(i):
        v = dict(fr=30.0, frame_count=counts[i], w=640, h=360, frames=resident[i])
        if mode != 'shot_net':
            v['trans_inds'] = cuts_of(i) + [counts[i]]
        return v

    t0 = time.perf_counter()
    js = scheduler.JobScheduler(CP, ratios, lanes=lanes, state_dict=sd, shot_net=shot_net)
    torch.cuda.synchronize()
    create_s = time.perf_counter() - t0
    runs, stats, allb = [], [], None
    n_runs = 5 if mode == 'resident' else 2

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for rep in range(n_runs):
        barrier()
        t0 = time.perf_counter()
        allb, st = svc_dist.crop_job(make, counts, ['%03d' % v for v in vids], CP, ratios, out_dir=None,
                                     crop_fn=lambda vs, cp, rs, w: js.run(vs))
        barrier()
        runs.append(time.perf_counter() - t0)
        stats.append(dict(js.stats))
    js.close()
    if shot_net is not None:
        shot_net.close()
    if mode == 'shot_net':
        n_sel_mine = int(stats[-1]['network_frames'])
    n_sel = torch.tensor([n_sel_mine], dtype=torch.int64,
                         device=dev if not dist_on or torch.distributed.get_backend() == 'nccl' else torch.device('cpu'))
    if dist_on:
        torch.distributed.all_reduce(n_sel)
    if rank != 0:
        return None
    annots = E.load_annotations(folder)
    boxes = {ar: {vids[i]: allb[r][i] for i in range(len(vids))} for ar, r in (('1-3', '1:3'), ('3-1', '3:1'))}
    gt, mt, index = E.pair_boxes(annots, boxes)
    scores = E.aggregate(np.asarray(ops.iou_boxes(gt, mt), np.float64), index)
    timed = sorted(runs[1:])
    best = timed[(len(timed) - 1) // 2]                       # the median of the runs behind the first (the lower middle one of an even count)
    k = runs.index(best)
    n_sel = int(n_sel.item())
    where = {'resident': 'selected frames resident in HBM',
             'host_fed': 'selected frames in PINNED HOST memory, copied and down-scaled inside the job (smartVidCrop._HostFeed): the clock includes PCIe',
             'shot_net': 'EVERY frame of every video resident in HBM, no trans_inds: TransNet V1 (synthetic weights, last layer biased to '
                         '"no transition": one shot per video) runs on the device inside the job and decides shots and selection (the '
                         'reference\'s video path)'}[mode]
    return dict(workload='200-video RetargetVid-shaped synthetic set (real frame counts%s), targets 1:3 and 3:1, %s; '
                         'dist.crop_job -> scheduler.JobScheduler -> all_gather of the boxes' % ('' if mode == 'shot_net' else ', 0-3 cuts per video', where),
                mode=mode, **({'shot_net_matrix_pipe': shot_pipe} if mode == 'shot_net' else {}),
                videos=len(vids), video_frames=int(sum(counts)), saliency_frames=n_sel, n_gpus=world, lanes_per_gpu=lanes,
                seconds=round(best, 4), seconds_all_runs=[round(r, 4) for r in runs],
                seconds_note='job wall clock between barriers (slowest rank), scheduler already created; the first run also pays '
                             'one-time allocations; `seconds` = the MEDIAN of the runs behind the first',
                video_frames_per_s=round(sum(counts) / best, 1), saliency_frames_per_s=round(n_sel / best, 1),
                per_rank_fixed_costs_s=dict(scheduler_create=round(create_s, 4), rccl_init=(None if rccl_init_s is None else round(rccl_init_s, 4)),
                                            generate_resident_frames=round(gen_s, 3)),
                scheduler_rank0={kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in stats[k].items()},
                windows_crc32=synth.windows_crc32(allb, ratios, len(vids)),
                eval_percent={ar: [round(x, 3) for x in sc] for ar, sc in scores.items()},
                eval_note='synthetic pixels against the human annotations: the scores only show that the job is deterministic')


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus N` (self-launching) or '
                         '`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`' % (args.gpus, world))
    # BENCH_SHARE_GPU=1 (test aid, never a measurement): the ranks share the visible GPUs (rank -> cuda:LOCAL_RANK % count) and
    # talk over gloo (RCCL refuses two ranks on one device), so that `bench.py --gpus 2` -- the self-launch, the sharding by
    # rank, the barriers, the MAX over ranks and the box gather -- runs on a one-GPU box (tests/test_gpu_dist_rccl.py)
    share_gpu = os.environ.get('BENCH_SHARE_GPU', '0') == '1'
    if args.gpus > torch.cuda.device_count() and not share_gpu:
        raise SystemExit('bench.py: --gpus %d but only %d GPU(s) visible' % (args.gpus, torch.cuda.device_count()))
    # BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, barrier, MAX all-reduce, box gather) with any
    # world size, including 1 under torch.distributed.run -- a way to exercise it on a single-GPU box
    dist_on = world > 1 or os.environ.get('BENCH_FORCE_DIST', '0') == '1'
    rccl_init_s = None
    if dist_on:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if share_gpu:
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
        t_init = time.perf_counter()
        if share_gpu:
            torch.distributed.init_process_group('gloo')
        else:
            torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))
        torch.distributed.barrier()                     # the communicator is built by the first collective
        rccl_init_s = time.perf_counter() - t_init
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())
    coll_dev = torch.device('cpu') if share_gpu else dev      # where the (tiny) collective payloads live: the device under RCCL

    B = args.batch
    P = max(1, args.pipeline)
    sd = weights.make_synthetic_state_dict(0)
    CP = S.sc_init_crop_params()
    CP['out_ratio'] = '1:3'
    frames_host = synth.blob_frames(B, 360, 640, seed=100 + rank, **BENCH_BLOBS)
    frames = torch.from_numpy(frames_host).to(dev)
    flags = np.zeros(B, np.uint8)
    if os.environ.get('BENCH_NO_BLEND', '0') != '1':   # (diagnostic: a batch without a cut has one tail round instead of three)
        flags[:2] = 1                  # the batch starts a shot: maps 0,1 blend into 1,2 (smartVidCrop.py:2369-2373)

    DEPTH = int(os.environ.get('BENCH_DEPTH', 2))              # calls a stream may have outstanding before the host collects the oldest
    spans = {'net': 0.0, 'tail': 0.0, 'n': 0}                  # device milliseconds of a batch's two phases (HIP events on its stream)
    host_t = {'wait': 0.0, 'boxes': 0.0, 'enqueue': 0.0}      # host seconds: waiting for a batch, boxes on the host, enqueueing a batch

    class Slot:
        """One in-flight step: its own engine (weights + workspace), HIP stream and pinned result buffer,
        so the low-occupancy clustering tail of one batch overlaps the network of the next.  enqueue / finish = the
        plain call (every blend chain run out in rounds inside the call); `pipe` = the product's streaming scheduler
        (retargetvid_amd/pipeline.py: one tail round per call, chains carried over to the stream's next calls)."""
        def __init__(self):
            self.eng = ops.Engine(sd)
            self.stream = lane_pool[len(slots_made)]        # the process's lane streams (scheduler.lane_streams): config 3 below re-uses them
            slots_made.append(self)
            self.xy_host = torch.empty((B, 2), dtype=torch.float64).pin_memory()
            self.maps = torch.empty((B, 140, 250), dtype=torch.uint8, device=dev)
            self.net_done = torch.cuda.Event(enable_timing=True)
            self.start = torch.cuda.Event(enable_timing=True)
            self.done = torch.cuda.Event(enable_timing=True)
            self.pending = False
            # every slot works on its OWN batch (same generator, another seed): the maps of the batches in flight then hold
            # different numbers of points (round-4 verdict: one batch fed every step of every slot, so every map had the same N);
            # slot 0 keeps the batch of rounds 1-4 (the un-pipelined measurements and the CPU baseline use it)
            k_ = len(slots_made) - 1
            self.frames = frames if k_ == 0 else torch.from_numpy(synth.blob_frames(B, 360, 640, seed=100 + rank + 1000 * k_, **BENCH_BLOBS)).to(dev)
            self.pipe = pipeline.StreamPipeline(self.eng, CP, 140, 250, batch=B, stream=self.stream, timing=True, depth=DEPTH)

        def enqueue(self):
            with torch.cuda.stream(self.stream):
                self.start.record(self.stream)
                small = self.eng.resize_frames(self.frames, 140, 250)
                maps = self.eng.saliency(small, out=self.maps)
                self.eng.threshold_(maps, CP['t_threshold'])
                self.net_done.record(self.stream)
                xy = self.eng.cluster_center_(maps, flags, CP)
                self.xy_host.copy_(xy, non_blocking=True)
                self.done.record(self.stream)
            self.pending = True

        def finish(self):
            t0 = time.perf_counter()
            self.done.synchronize()
            t1 = time.perf_counter()
            self.pending = False
            spans['net'] += self.start.elapsed_time(self.net_done)
            spans['tail'] += self.net_done.elapsed_time(self.done)
            spans['n'] += 1
            b = host_boxes(self.xy_host.numpy())
            host_t['wait'] += t1 - t0
            host_t['boxes'] += time.perf_counter() - t1
            return b

    from retargetvid_amd import scheduler as _sched
    lane_pool, slots_made = _sched.lane_streams(dev, P), []
    slots = [Slot() for _ in range(P)]
    torch.cuda.synchronize()
    STREAMED = os.environ.get('BENCH_CARRY', '1') != '0' and P > 1      # BENCH_CARRY=0: the plain call (rounds inside the call)

    def run_streamed(steps):
        """The timed path: every slot's StreamPipeline gets a batch per turn; a batch's boxes are computed when its last
        centre has arrived (the maps of its blend chain finish in the stream's next calls); finish() completes every
        batch inside the timed region."""
        boxes = None
        pend = {}                                             # (slot, batch number on its stream) -> [centres, count]
        for sl in slots:
            sl.pipe.reset()

        def take(k, results):
            nonlocal boxes
            t1 = time.perf_counter()
            for g, x, y in results:
                ent = pend.setdefault((k, g // B), [np.empty((B, 2), np.float64), 0])
                ent[0][g % B] = (x, y)
                ent[1] += 1
                if ent[1] == B:
                    boxes = host_boxes(ent[0])
                    del pend[(k, g // B)]
            host_t['boxes'] += time.perf_counter() - t1

        for s in range(steps):
            k = s % P
            sl = slots[k]
            if len(sl.pipe.calls) >= sl.pipe.depth:
                t0 = time.perf_counter()
                res = sl.pipe.collect()
                host_t['wait'] += time.perf_counter() - t0
                take(k, res)
            t0 = time.perf_counter()
            sl.pipe.submit_frames(sl.frames, flags)
            host_t['enqueue'] += time.perf_counter() - t0
        for k, sl in enumerate(slots):                         # every stream's last call is enqueued before any is waited for
            take(k, sl.pipe.flush())
        for k, sl in enumerate(slots):
            t0 = time.perf_counter()
            res = sl.pipe.finish()
            host_t['wait'] += time.perf_counter() - t0
            take(k, res)
        assert not pend, 'every batch must be complete at the end of the timed region'
        for sl in slots:
            net, tail = sl.pipe.phase_ms()
            spans['net'] += net; spans['tail'] += tail; spans['n'] += 1
        return boxes

    def run(steps):
        if STREAMED:
            return run_streamed(steps)
        boxes = None
        for s in range(steps):
            sl = slots[s % P]
            if sl.pending:
                boxes = sl.finish()
            t0 = time.perf_counter()
            sl.enqueue()
            host_t['enqueue'] += time.perf_counter() - t0
        for sl in slots:
            if sl.pending:
                boxes = sl.finish()
        return boxes

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    boxes = run(max(args.warmup, P))
    # the timed path against the plain call on one batch (cheap; the full statement is tests/test_gpu_pipeline.py): the
    # streaming scheduler must give the centres of the call that runs every blend chain out in rounds
    if STREAMED:
        slots[0].enqueue()
        slots[0].finish()
        xy_plain = slots[0].xy_host.numpy().copy()
        slots[0].pipe.reset()
        slots[0].pipe.submit_frames(frames, flags)
        xy_stream = np.full((B, 2), np.nan)
        for g, x, y in slots[0].pipe.finish():
            xy_stream[g] = (x, y)
        if not np.array_equal(xy_plain, xy_stream, equal_nan=True):
            raise SystemExit('bench.py: the streaming schedule and the plain call disagree on the centres of one batch')
    # 1. un-pipelined pass (one batch in flight, slot 0 only): per kernel class, iso_steps profiled steps -> launch
    #    durations that overlap nothing; also the latency of one batch and the step time with one batch in flight
    eng = slots[0].eng
    iso = max(1, args.iso_steps)
    per_class, raw_class = {}, {}
    plain = os.environ.get('BENCH_PLAIN', '0') == '1'            # counter passes (tools/refresh_profiles.sh): only warm-up + timed steps run
    for k in eng.KERNEL_CLASSES:
        if plain:
            per_class[k] = (1.0 if k == 'pw' else 0.0, 1.0)
            raw_class[k] = (per_class[k][0], 0.0)
            continue
        eng.profile_enable(k)
        for _ in range(iso):
            slots[0].enqueue()
            slots[0].finish()
        raw, pair, cnt = eng.profile_read_raw()
        per_class[k] = (raw / iso, cnt / iso)                     # ms per step (the raw event-pair durations), launches per step
        raw_class[k] = (max(0.0, raw - 0.75 * pair * cnt) / iso, pair)       # minus 3/4 of an empty event pair per launch: secondary key
    eng.profile_enable(None)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(0 if plain else 10):
        slots[0].enqueue()
        slots[0].finish()
    latency_ms = None if plain else (time.perf_counter() - t1) / 10 * 1e3
    # points per map of this workload (what the clustering kernels see)
    if plain:
        npts = np.zeros(B, np.int64)                              # (counter passes: nothing but warm-up + timed steps may run)
    else:
        with torch.cuda.stream(slots[0].stream):
            npl = []
            for sl_ in slots:                                         # every slot's batch
                m_ = eng.saliency(eng.resize_frames(sl_.frames, 140, 250))
                eng.threshold_(m_, CP['t_threshold'])
                _, st_ = eng.cluster_center_(m_, flags, CP, want_stats=True)
                npl.append(st_[:, 0].cpu().numpy())
        npts = np.concatenate(npl)
    dominant = max(per_class, key=lambda k: per_class[k][0])
    # BENCH_LIVE_PROFILE=1 also records the dominant class's events INSIDE the timed region (roofline.*_in_flight).  Off by
    # default: 76 event records per step on every stream cost 2-3 % of the step with four batches in flight (1.49 -> 1.45 ms)
    # and more with deeper pipelines; roofline.frac comes from the un-pipelined passes above either way.
    live = os.environ.get('BENCH_LIVE_PROFILE', '0') == '1' and not plain
    for sl in slots:
        sl.eng.profile_enable(dominant if live else None)
        sl.eng.profile_read()

    # 2. the timed region: K steps, P batches in flight; repeated, the median region is the one reported
    regions = []
    for rep in range(max(1, args.repeats)):
        barrier()
        for k in host_t:
            host_t[k] = 0.0
        spans.update(net=0.0, tail=0.0, n=0)
        t0 = time.perf_counter()
        boxes = run(args.steps)
        barrier()
        dt_r = time.perf_counter() - t0
        regions.append((dt_r, {k: round(v / max(args.steps, 1) * 1e3, 4) for k, v in host_t.items()},
                        {'network': round(spans['net'] / max(spans['n'], 1), 4), 'tail': round(spans['tail'] / max(spans['n'], 1), 4)}))
    order_r = sorted(range(len(regions)), key=lambda i: regions[i][0])
    dt_local, host_ms, span_ms = regions[order_r[len(order_r) // 2]]
    dt = dt_local
    fl_ms, fl_launches = 0.0, 0
    for sl in slots:
        ms, cnt = sl.eng.profile_read()
        fl_ms += ms
        fl_launches += cnt
        sl.eng.profile_enable(None)

    rank_fps = [B * args.steps / dt_local]
    seen_world = 1
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
        seen_world = torch.distributed.get_world_size()
        tl = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(seen_world)]
        torch.distributed.all_gather(tl, torch.tensor([dt_local], dtype=torch.float64, device=coll_dev))
        rank_fps = [B * args.steps / float(t.item()) for t in tl]

    # rounds 1-4 fed ONE batch to every slot (every map of a step then held the same number of points): the same timed region that
    # way, three repeats, for comparison across rounds (config.one_batch_for_every_slot)
    same_batch = None
    if not plain and P > 1:
        own = [sl.frames for sl in slots]
        for sl in slots:
            sl.frames = frames
        sb = []
        for rep in range(3):
            barrier()
            t0 = time.perf_counter()
            run(args.steps)
            barrier()
            sb.append((time.perf_counter() - t0) / max(args.steps, 1) * 1e3)
        for sl, fr_ in zip(slots, own):
            sl.frames = fr_
        sbm = sorted(sb)[1]
        same_batch = dict(ms_per_step=round(sbm, 4), frames_per_s=round(world * B / sbm * 1e3, 1), ms_per_step_all=[round(v, 4) for v in sb],
                          note='every slot on slot 0\'s batch, as in rounds 1-4 (1 640 - 2 057 points per map)')
    # the OTHER matrix pipe (SVC_MX=f32 | bf16x6), timed beside the default in the same run: the same timed region on engines created
    # with it, five repeats, median.  Reported under config.matrix_pipe_variant, never as `value`.
    variant = None
    if os.environ.get('BENCH_VARIANT', '1') != '0' and not plain:
        other_pipe = 'f32' if eng.matrix_pipe() == 'bf16x6' else 'bf16x6'
        old_mx = os.environ.get('SVC_MX')
        try:
            os.environ['SVC_MX'] = other_pipe
            slots_made.clear()
            f32_slots = list(slots)
            slots[:] = [Slot() for _ in range(P)]
            if old_mx is None:
                os.environ.pop('SVC_MX', None)
            else:
                os.environ['SVC_MX'] = old_mx
            run(max(args.warmup, P))
            vr = []
            for rep in range(5):
                barrier()
                t0 = time.perf_counter()
                run(args.steps)
                barrier()
                vr.append((time.perf_counter() - t0) / max(args.steps, 1) * 1e3)
            vms = sorted(vr)[len(vr) // 2]
            variant = dict(matrix_pipe=slots[0].eng.matrix_pipe(), ms_per_step=round(vms, 4), frames_per_s=round(world * B / vms * 1e3, 1),
                           ms_per_step_all=[round(v, 4) for v in vr],
                           note='the timed region on engines created with SVC_MX=%s (bf16x6 = the 1x1 convolutions on '
                                'v_mfma_f32_32x32x16_bf16 with every f32 operand split into three bf16 planes: 24 significant bits, six plane '
                                'pairs per product, f32 accumulation; f32 = v_mfma_f32_32x32x2_f32); DESIGN.md 5' % other_pipe)
            for sl in slots:
                sl.eng.close()
            slots[:] = f32_slots
        except Exception as e:
            if old_mx is None:
                os.environ.pop('SVC_MX', None)
            else:
                os.environ['SVC_MX'] = old_mx
            variant = dict(error=repr(e))
    c3 = c3_host = c3_shot = c3_shot3 = None
    if os.environ.get('BENCH_CONFIG3', '1') != '0' and not plain:
        c3_lanes = int(os.environ.get('BENCH_CONFIG3_LANES', 12))
        try:
            c3 = config3_job(world, rank, dist_on, dev, sd, c3_lanes, rccl_init_s)
        except Exception as e:                                 # the headline line must not depend on the extra job
            c3 = dict(error=repr(e))
        # the two system-level variants (N = 1 only: 14.5 GB of pinned host memory / 85 GB of HBM per job): BENCH_CONFIG3_EXTRA=0 skips them
        if world == 1 and os.environ.get('BENCH_CONFIG3_EXTRA', '1') != '0':
            # shot_net twice: on the library's default pipe (bf16x6: fp32-class probabilities) and on the opt-in three-pair form
            for mode, smx in (('host_fed', None), ('shot_net', None), ('shot_net', 'bf16x3')):
                torch.cuda.empty_cache()
                try:
                    import psutil
                    if mode == 'host_fed' and psutil.virtual_memory().available < (40 << 30):
                        raise RuntimeError('less than 40 GB of host memory available: not pinning 14.5 GB')
                    r_ = config3_job(world, rank, dist_on, dev, sd, c3_lanes, rccl_init_s, mode=mode, shot_mx=smx)
                except Exception as e:
                    r_ = dict(error=repr(e))
                if mode == 'host_fed':
                    c3_host = r_
                elif smx is None:
                    c3_shot = r_
                else:
                    c3_shot3 = r_
            torch.cuda.empty_cache()
    if rank == 0:
        front_fused = eng.front_fused()
        work = layer_work(B, front_fused=front_fused)
        steps = max(args.steps, 1)
        iso_ms, iso_n = per_class[dominant]
        byte_work = {'pw': work['pw_bytes'], 'dw': work['dw_bytes'], 'lanczos': B * (140 * 250 * 3 + 256 * 416 * 12.0),
                     'stem': B * (256 * 416 * 12.0 + 128 * 208 * 128.0), 'resize': B * (360 * 640 * 3 + 140 * 250 * 3.0),
                     'smooth': B * (32 * 52 * 4 + 140 * 250 * 9.0)}.get(dominant, work['map_bytes'])
        if dominant == 'pw':
            unit_work, peak, unit, bound, scale = work['pw_flops'], MFMA_F32_PEAK_TFLOPS, 'TFLOP/s', 'mfma', 1e12
        else:
            unit_work, peak, unit, bound, scale = byte_work, HBM_PEAK_GBS, 'GB/s', 'hbm', 1e9
        rate = lambda ms_per_step: unit_work / (ms_per_step * 1e-3) / scale if ms_per_step > 0 else 0.0
        ach = rate(iso_ms)
        roof = dict(bound=bound, kernel=dominant, achieved=round(ach, 3), peak=peak, unit=unit, frac=round(ach / peak, 5),
                    measured='HIP events around every launch of the class in %d un-pipelined steps (one batch in flight) '
                             'before the timed region' % iso,
                    launches_per_step=round(iso_n, 2), class_ms_per_step=round(iso_ms, 4),
                    class_ms_per_step_event_corrected=round(raw_class[dominant][0], 4), empty_event_pair_ms=round(raw_class[dominant][1], 5),
                    frac_event_corrected=round(rate(raw_class[dominant][0]) / peak, 5) if raw_class[dominant][0] > 0 else None,
                    event_correction='frac / achieved / class_ms_per_step are the RAW event-pair durations (they include the events\' own '
                                     'cost, so they read a few percent below a profiler\'s kernel durations); *_event_corrected takes 0.75 x '
                                     'an empty event pair off every launch',
                    avg_launch_ms=round(iso_ms / max(iso_n, 1), 5),
                    algorithmic_per_step=unit_work, algorithmic_unit='FLOP' if dominant == 'pw' else 'B',
                    frac_wall=round(rate(dt / steps * 1e3) / peak, 5))
        if live and fl_launches:
            fl = rate(fl_ms / steps)
            roof.update(achieved_in_flight=round(fl, 3), frac_in_flight=round(fl / peak, 5),
                        avg_launch_ms_in_flight=round(fl_ms / fl_launches, 5),
                        in_flight_note='timed region, %d batches in flight: launch durations overlap other streams\' kernels, '
                                       'their sum (%.3f ms per step) exceeds the wall clock' % (P, fl_ms / steps))
        if dominant == 'pw':
            gbs = work['pw_bytes'] / (iso_ms * 1e-3) / 1e9
            roof.update(hbm_achieved_GBs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 5),
                        hbm_note='un-fused layer-wise fp32 traffic of the class (in + out + weights) / class time')
        roof['traffic'] = None
        try:
            src = next(os.path.join('profiles', f) for f in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json')
                       if os.path.isfile(os.path.join(ROOT, 'profiles', f)))
            with open(os.path.join(ROOT, src)) as fp:
                c = json.load(fp)['classes'][dominant]
            roof.update(traffic=c['hbm_bytes_per_launch'], traffic_per_step=c.get('hbm_bytes_per_step'),
                        traffic_source='%s (stored rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --pipeline 1, '
                                       'reduced by tools/pmc_traffic.py; not re-measured in this run)' % src)
        except Exception:
            pass
        roof['class_ms_per_step_all'] = {k: round(v[0], 4) for k, v in per_class.items()}
        net_classes = [k for k in per_class if k not in ('compact', 'core', 'prim', 'finish')]
        roof['device_ms_per_unpipelined_step'] = dict(
            network=round(sum(per_class[k][0] for k in net_classes), 4), tail=round(sum(per_class[k][0] for k in ('compact', 'core', 'prim', 'finish')), 4),
            note='summed launch durations of one un-pipelined step: the tail (one 1024-thread workgroup per map: 32 of 256 CUs) is the '
                 'larger DEVICE time and the smaller share of the chip; `kernel` above is the largest single class')
        # which matrix pipe the class ran on (svc_matrix_pipe): with split-bf16 operands every f32 operand is three bf16 planes
        # (24 significant bits) and six plane pairs per product are accumulated in f32 -- `frac` stays the ALGORITHMIC f32 FLOPs
        # against the fp32-MFMA roof (it may now exceed what that pipe could do), the bf16 roof is quoted beside it
        mp = eng.matrix_pipe()
        roof['matrix_pipe'] = mp
        if dominant == 'pw':
            roof['peak_bf16_TFLOPs'] = MFMA_BF16_PEAK_TFLOPS
            roof['frac_vs_bf16_roof'] = round(ach / MFMA_BF16_PEAK_TFLOPS, 5)
            if mp == 'bf16x6':
                roof['frac_vs_bf16_roof_executed'] = round(6.0 * ach / MFMA_BF16_PEAK_TFLOPS, 5)
                roof['matrix_pipe_note'] = ('bf16x6 (the default since round 5): v_mfma_f32_32x32x16_bf16 on operands split into three bf16 planes (8 + 8 + 8 = 24 '
                                            'significant bits), six plane pairs per product, f32 accumulation; the expand GEMM of fused blocks 2, 3, '
                                            '5-7 and every un-fused 1x1 convolution run this way, block 4 and the project GEMMs of blocks 2-6 on the '
                                            'fp32 MFMA; executed = 6 x algorithmic FLOPs')
            else:
                roof['matrix_pipe_note'] = ('f32 (SVC_MX=f32): v_mfma_f32_32x32x2_f32 / 16x16x4 (exact f32 products); the split-bf16 form is measured '
                                            'beside it as config.matrix_pipe_variant')
        if front_fused:
            roof['class_note'] = ("'stem' = k_front: LANCZOS + features.0 + features.1 in one kernel; features.1's project is not in "
                                  "the 'pw' FLOPs")
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            torch.set_num_threads(min(16, os.cpu_count() or 1))    # batch-1 convs stop scaling (and collapse) beyond this
            nb = max(1, args.cpu_sample // B)
            secs, nfr = 0.0, 0
            for b in range(nb):                                    # whole steps of the same workload, new frames each
                fh = frames_host if b == 0 else synth.blob_frames(B, 360, 640, seed=200 + b, **BENCH_BLOBS)
                fps_b, s_b = cpu_baseline(sd, fh, CP, flags)
                secs += s_b
                nfr += B
            cpu = dict(value=round(nfr / secs, 3), unit='frames/s', cores=torch.get_num_threads(), host_cpu_count=os.cpu_count(), kind='port',
                       sample='%d steps of the same workload (%d frames), oracle/ (PyTorch-CPU fp32 forward at batch 1, '
                              'NumPy tail), %.1f s' % (nb, nfr, secs))
            if os.environ.get('BENCH_TORCH_BASELINE', '1') != '0':  # the reference's own way of running the network, on this GPU
                try:
                    cpu['network_under_pytorch_rocm'] = reference_formulation_on_gpu(sd, dev, B)
                except Exception as e:                              # (MIOpen needs a writable cache directory: a baseline must not fail the bench)
                    cpu['network_under_pytorch_rocm'] = dict(error='%s: %s' % (type(e).__name__, str(e)[:200]))
        value = world * B * args.steps / dt
        tail_classes = {k: round(per_class[k][0], 4) for k in ('compact', 'core', 'prim', 'finish')}
        roof['tail'] = dict(total_ms=round(sum(tail_classes.values()), 4), prim_ms=tail_classes['prim'], finish_ms=tail_classes['finish'], core_ms=tail_classes['core'],
                            compact_ms=tail_classes['compact'],
                            points_per_map=dict(min=int(npts.min()), mean=round(float(npts.mean()), 1), max=int(npts.max())),
                            us_per_frame=round(sum(tail_classes.values()) / B * 1e3, 2),
                            note='class ms per un-pipelined step of %d maps (three tail rounds: the batch starts a shot).  Default: a '
                                 'round is two fused launches -- prim = k_tail_front (compact + core + Prim), finish = k_tail_back '
                                 '(sort + hierarchy + finish), compact / core then read 0; SVC_TAIL_MERGE=0: one launch per stage, '
                                 'finish = k_sort + k_tree_par + k_finish' % B)
        out = dict(metric='frames/sec end-to-end saliency+crop on 640x360', value=round(value, 2), unit='frames/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / max(args.steps, 1) * 1e3, 4),
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='Single 640x360 video, batch=32 frames, UNISAL saliency + crop on 1 MI355X',
                               workload_id=('r02-1to3blobs-sigma20-60' if not BENCH_BLOBS else
                                            'r03-%dblobs-sigma%g-%g' % (BENCH_BLOBS['n_blobs'], BENCH_BLOBS['sigma'][0], BENCH_BLOBS['sigma'][1])),
                               config3=c3, config3_host_fed=c3_host, config3_shot_net=c3_shot, config3_shot_net_bf16x3=c3_shot3, matrix_pipe_variant=variant,
                               one_batch_for_every_slot=same_batch,
                               batch_per_gpu=B, frame='640x360x3 u8', saliency_map='140x250 u8', network_input='256x416',
                               weights='synthetic seed 0 (weights.make_synthetic_state_dict)',
                               video_frames_per_s=round(value * CP['skip'], 1), batches_in_flight=P,
                               distinct_batches='every slot steps through its own batch of 32 frames (same generator, seeds 100 + 1000 k); rounds 1-4 fed one batch to every slot',
                               parallelism='frames sharded, dp%d' % world,
                               world_size_seen_by_rccl=seen_world if dist_on else None,
                               ranks_share_gpus=('BENCH_SHARE_GPU=1: %d ranks on %d visible GPU(s) over gloo -- a functional run of the multi-rank '
                                                 'path, NOT a measurement' % (world, torch.cuda.device_count())) if share_gpu else None,
                               per_rank_frames_per_s=[round(v, 1) for v in rank_fps],
                               points_per_map=dict(min=int(npts.min()), mean=round(float(npts.mean()), 1), max=int(npts.max())),
                               repeats=dict(regions=len(regions), ms_per_step=[round(r[0] / max(args.steps, 1) * 1e3, 4) for r in regions],
                                            reported='median'),
                               one_batch_in_flight=(None if latency_ms is None else
                                                    dict(latency_ms_per_batch=round(latency_ms, 4),
                                                         frames_per_s=round(B / latency_ms * 1e3, 1))),
                               host_ms_per_step=host_ms, batch_phase_ms_in_the_pipeline=span_ms,
                               blend_chain=('the chains run out in rounds inside every call' if not STREAMED else
                                            'retargetvid_amd.pipeline.StreamPipeline: the maps behind a cut are clustered in round 0 of '
                                            'the next calls on their stream (SVC_MAP_HELD): one tail round per call')),
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
