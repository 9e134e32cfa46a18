"""Run every HIP-vs-oracle comparison and print a diagnostic summary (GPU box helper)."""
import os, sys, time, traceback
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, weights, synth
from oracle import unisal_ref as U, cv_ref, tail_ref as T, hdbscan_ref as H, pipeline_ref as P

torch.set_num_threads(8)
os.environ.setdefault('SVC_KEEP_INPUT', '1')      # the input tap below needs the fused front kernel to write it
sd = weights.make_synthetic_state_dict(0)
eng = ops.Engine(sd)
ok = True

def section(name):
    print('\n=== %s ===' % name, flush=True)

try:
    section('resize')
    fr = synth.blob_frames(3, 360, 640, seed=5)
    got = eng.resize_frames(torch.from_numpy(fr).cuda(), 140, 250).cpu().numpy()
    ref = np.stack([cv_ref.resize_linear_u8(f, 140, 250) for f in fr])
    print('mismatch', (got != ref).sum(), 'max', np.abs(got.astype(int) - ref).max())
    ok &= (got == ref).all()
except Exception:
    traceback.print_exc(); ok = False

try:
    section('saliency')
    g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'unisal_golden.npz'))
    frames = g['frames']
    t0 = time.time()
    maps = eng.saliency(torch.from_numpy(frames).cuda()); torch.cuda.synchronize()
    print('first call %.3fs' % (time.time() - t0))
    maps = maps.cpu().numpy()
    taps = {}
    ref_maps = U.saliency_u8(sd, frames, taps)          # [h,w,n]
    t = taps['frames'][0]
    def cmp(name, got, ref):
        d = np.abs(got - ref)
        print('%-10s max|d| %.3e  rel %.3e  (ref absmax %.3e)' % (name, d.max(), d.max() / (np.abs(ref).max() + 1e-30), np.abs(ref).max()))
        return d.max() / (np.abs(ref).max() + 1e-30)
    r = cmp('input', eng.tap(ops.TAP_INPUT, 0, (256, 416, 3)), t['input'][0].permute(1, 2, 0).numpy())
    ok &= r == 0
    ok &= cmp('feat_4x', eng.tap(ops.TAP_FEAT4X, 0, (32, 52, 64)), t['feat_4x'][0].permute(1, 2, 0).numpy()) < 1e-4
    ok &= cmp('feat_2x', eng.tap(ops.TAP_FEAT2X, 0, (16, 26, 160)), t['feat_2x'][0].permute(1, 2, 0).numpy()) < 1e-4
    f1 = eng.tap(ops.TAP_FEAT1X, 0, (8, 13, 1296))
    ok &= cmp('feat_1x', f1[:, :, :1280], t['feat_1x'][0].permute(1, 2, 0).numpy()) < 1e-4
    gm = U.gaussian_maps(torch.from_numpy(sd['coarse_gaussians_salicon']), 8, 13).permute(1, 2, 0).numpy()
    ok &= cmp('gauss', f1[:, :, 1280:], gm) < 1e-5
    ok &= cmp('post_cnn', eng.tap(ops.TAP_POSTCNN, 0, (8, 13, 256)), t['post_cnn'][0].permute(1, 2, 0).numpy()) < 1e-4
    ok &= cmp('dec', eng.tap(ops.TAP_DEC, 0, (32, 52, 64)), t['dec'][0].permute(1, 2, 0).numpy()) < 1e-4
    ok &= cmp('pre', eng.tap(ops.TAP_PRE, 0, (140, 250)), t['pre'][0].numpy()) < 1e-4
    for i in range(frames.shape[0]):
        d = np.abs(maps[i].astype(int) - ref_maps[:, :, i].astype(int))
        print('frame %d u8: mismatch %d (%.4f%%) max %d' % (i, (d > 0).sum(), 100.0 * (d > 0).mean(), d.max()))
        ok &= d.max() <= 1 and (d > 0).mean() < 0.01
    print('golden (reference model) mismatch', (np.transpose(maps, (1, 2, 0)) != g['smaps_u8']).sum())
except Exception:
    traceback.print_exc(); ok = False

try:
    section('tail')
    CP = P.init_crop_params()
    g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'unisal_golden.npz'))
    base = np.transpose(g['smaps_u8'], (2, 0, 1)).copy()              # [4,h,w]
    # extra synthetic maps: speckle, empty, tiny
    rng = np.random.RandomState(3)
    extra = []
    m = np.zeros((140, 250), np.uint8); extra.append(m)               # empty
    m = np.zeros((140, 250), np.uint8); m[10:13, 10:14] = 200; extra.append(m)   # 12 points < mcs
    m = (rng.rand(140, 250) < 0.03).astype(np.uint8) * rng.randint(120, 256, (140, 250)).astype(np.uint8); m[50:80, 100:160] = 250; extra.append(m)
    m = base[1].copy(); m[m < 60] = 0; extra.append(m)                # big blobs
    maps = np.concatenate([base, np.stack(extra)]).copy()
    n = maps.shape[0]
    flags = np.zeros(n, np.uint8); flags[0] = 1; flags[1] = 1; flags[5] = 1       # chains 0->1->2, 5->6
    ref = maps.copy()
    T.threshold(ref, CP['t_threshold'])
    ref_hwn = np.ascontiguousarray(np.transpose(ref, (1, 2, 0)))
    infos = []
    for i in range(n):
        info = {}
        ref_hwn[:, :, i] = T.clustering_filt(ref_hwn[:, :, i], CP, info); infos.append(info)
        if i + 1 < n and flags[i]:
            ref_hwn[:, :, i + 1] = T.blend_next(ref_hwn[:, :, i], ref_hwn[:, :, i + 1])
    dx, dy = T.centers(ref_hwn, CP)
    dm = torch.from_numpy(maps).cuda()
    eng.threshold_(dm, CP['t_threshold'])
    t0 = time.time()
    xy, stats = eng.cluster_center_(dm, flags, CP, want_stats=True); torch.cuda.synchronize()
    print('cluster_center %.3fs' % (time.time() - t0))
    got = dm.cpu().numpy(); xy = xy.cpu().numpy(); stats = stats.cpu().numpy()
    for i in range(n):
        mm = (got[i] != ref_hwn[:, :, i]).sum()
        rx = (np.nan, np.nan) if dx[i] is None else (dx[i], dy[i])
        same_xy = (np.isnan(xy[i, 0]) and dx[i] is None) or (dx[i] is not None and xy[i, 0] == dx[i] and xy[i, 1] == dy[i])
        print('map %d: N=%s stats=%s map mismatch %d  xy gpu (%.6f,%.6f) ref (%.6f,%.6f) %s' % (
            i, infos[i].get('n_points'), stats[i].tolist(), mm, xy[i, 0], xy[i, 1], rx[0], rx[1], 'OK' if same_xy and mm == 0 else 'DIFF'))
        ok &= bool(same_xy and mm == 0)
        st = eng.cluster_state(i, 35000)
        if 'labels' in infos[i] and st['n'] == infos[i]['n_points']:
            X = np.stack([st['pts'] & 255, (st['pts'] >> 8) & 255], 1).astype(np.int64)
            k = H.effective_min_samples(len(X), CP['hdbscan_min'], CP['hdbscan_min_samples'])
            core = H.core_distances(X, k)
            u, v, w = H.prim_mst(X, core)
            print('    core mismatch %d  mst mismatch %d  finish stamps (us) %s prim %.1f us nclusters %d' % (
                (core != st['core']).sum(), (np.stack([u, v, w], 1) != st['mst']).sum(), (st['hdr'][8:12] / 100.0).tolist(), st['hdr'][12] / 100.0, st['hdr'][4]))
except Exception:
    traceback.print_exc(); ok = False

try:
    section('iou')
    rng = np.random.RandomState(0)
    a = rng.randint(0, 600, (1000, 4)).astype(np.int32); a[:, 2:] += a[:, :2]
    b = rng.randint(0, 600, (1000, 4)).astype(np.int32); b[:, 2:] += b[:, :2]
    got = ops.iou_boxes(a, b)
    ref = np.array([T.iou(x, y) for x, y in zip(a.tolist(), b.tolist())])
    print('iou mismatch', (got != ref).sum())
    ok &= (got == ref).all()
except Exception:
    traceback.print_exc(); ok = False

try:
    section('timing')
    fr = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=0)).cuda()
    for k in eng.KERNEL_CLASSES:
        eng.profile_enable(k)
        small = eng.resize_frames(fr, 140, 250); maps = eng.saliency(small); eng.threshold_(maps, 120)
        eng.cluster_center_(maps, None, CP)
        print('  class %-10s %8.3f ms in %d launches' % ((k,) + eng.profile_read()))
    eng.profile_enable(None)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        small = eng.resize_frames(fr, 140, 250)
        torch.cuda.synchronize(); t1 = time.time()
        maps = eng.saliency(small)
        torch.cuda.synchronize(); t2 = time.time()
        eng.threshold_(maps, 120)
        xy, stats = eng.cluster_center_(maps, None, CP, want_stats=True)
        torch.cuda.synchronize(); t3 = time.time()
        print('B=32: resize %.2f ms, saliency %.2f ms, tail %.2f ms  (N mean %.0f)' % (
            (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, stats[:, 0].float().mean().item()))
    core_end = [eng.cluster_state(i, 8)['hdr'][5:8] / 100.0 for i in range(32)]
    worst = int(np.argmax([c[2] for c in core_end]))
    print('  k_core stamps (us): median phase1 %.1f phase2 %.1f end %.1f | slowest map %d: %.1f %.1f %.1f' % (
        np.median([c[0] for c in core_end]), np.median([c[1] for c in core_end]), np.median([c[2] for c in core_end]), worst, *core_end[worst]))
    for i in range(0, 32, 4):
        st = eng.cluster_state(i, 35000)
        print('  warm frame %2d: N=%5d nclusters %3d  finish stamps (us): sorted %.1f built %.1f hierarchy %.1f chosen %.1f done %.1f | prim %.1f us | k_tree shader clock %.0f MHz' % (
            i, st['n'], st['hdr'][4], st['hdr'][8] / 100.0, st['hdr'][13] / 100.0, st['hdr'][9] / 100.0, st['hdr'][10] / 100.0, st['hdr'][11] / 100.0, st['hdr'][12] / 100.0, st['hdr'][14] / max(st['hdr'][15], 1) * 100.0))
except Exception:
    traceback.print_exc(); ok = False

print('\nALL OK' if ok else '\nSOME CHECKS FAILED')
