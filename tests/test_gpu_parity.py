"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical inputs.
Integer / byte stages must match bit for bit; the fp32 network within the stated tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import cv_ref, hdbscan_ref as H, pipeline_ref as P, tail_ref as T, unisal_ref as U
from retargetvid_amd import ops, synth

pytestmark = pytest.mark.gpu


def _ref_tail(maps, flags, CP):
    """Oracle: threshold -> (filter, blend) loop -> centres on [n,h,w] u8."""
    ref = maps.copy()
    T.threshold(ref, CP['t_threshold'])
    hwn = np.ascontiguousarray(np.transpose(ref, (1, 2, 0)))
    npts = []
    for i in range(hwn.shape[2]):
        info = {}
        hwn[:, :, i] = T.clustering_filt(hwn[:, :, i], CP, info) if CP['clust_filt'] else hwn[:, :, i]
        npts.append(info.get('n_points', 0))
        if i + 1 < hwn.shape[2] and flags is not None and flags[i]:
            hwn[:, :, i + 1] = T.blend_next(hwn[:, :, i], hwn[:, :, i + 1])
    dx, dy = T.centers(hwn, CP)
    return np.transpose(hwn, (2, 0, 1)), dx, dy, npts


def _check_tail(engine, maps, flags, CP):
    ref_maps, dx, dy, _ = _ref_tail(maps, flags, CP)
    dm = torch.from_numpy(maps.copy()).cuda()
    engine.threshold_(dm, CP['t_threshold'])
    xy, stats = engine.cluster_center_(dm, flags, CP, want_stats=True)
    got, xy = dm.cpu().numpy(), xy.cpu().numpy()
    assert np.array_equal(got, ref_maps)
    for i in range(maps.shape[0]):
        if dx[i] is None:
            assert np.isnan(xy[i]).all()
        else:
            assert xy[i, 0] == dx[i] and xy[i, 1] == dy[i]          # equal to the ORACLE's centre (integer sums / count in float64); the reference's KMeans(n_clusters=1) agrees with that mean within 1e-12 px, not bit for bit (tests/test_oracle_tail.py)
    return stats.cpu().numpy()


def test_ingest_resize_bit_exact(engine):
    for (h, w, sh, sw) in [(360, 640, 140, 250), (1080, 1920, 140, 250), (480, 640, 187, 250), (37, 53, 20, 31)]:
        fr = np.random.RandomState(h).randint(0, 256, (2, h, w, 3)).astype(np.uint8)
        got = engine.resize_frames(torch.from_numpy(fr).cuda(), sh, sw).cpu().numpy()
        ref = np.stack([cv_ref.resize_linear_u8(f, sh, sw) for f in fr])
        assert np.array_equal(got, ref)


def test_saliency_against_oracle_and_reference_golden(engine, synthetic_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, 'unisal_golden.npz'))
    frames = g['frames']
    maps = engine.saliency(torch.from_numpy(frames).cuda()).cpu().numpy()
    taps = {}
    ref = U.saliency_u8(synthetic_sd, frames, taps)
    t = taps['frames'][0]
    assert np.array_equal(engine.tap(ops.TAP_INPUT, 0, (256, 416, 3)), t['input'][0].permute(1, 2, 0).numpy())   # K0 bit-exact
    # fp32 tolerance: 2e-4 of the tensor's max magnitude (accumulation order differs from the CPU)
    for which, shape, key in [(ops.TAP_FEAT4X, (32, 52, 64), 'feat_4x'), (ops.TAP_FEAT2X, (16, 26, 160), 'feat_2x'),
                              (ops.TAP_POSTCNN, (8, 13, 256), 'post_cnn'), (ops.TAP_DEC, (32, 52, 64), 'dec')]:
        r = t[key][0].permute(1, 2, 0).numpy()
        assert np.abs(engine.tap(which, 0, shape) - r).max() <= 2e-4 * np.abs(r).max(), key
    f1 = engine.tap(ops.TAP_FEAT1X, 0, (8, 13, 1296))
    r = t['feat_1x'][0].permute(1, 2, 0).numpy()
    assert np.abs(f1[:, :, :1280] - r).max() <= 2e-4 * np.abs(r).max()
    assert np.abs(engine.tap(ops.TAP_PRE, 0, (140, 250)) - t['pre'][0].numpy()).max() < 1e-4
    # u8 maps: |diff| <= 1 on < 0.1 % of the pixels, against the oracle AND the reference model's own output
    for r8 in (ref, g['smaps_u8']):
        d = np.abs(np.transpose(maps, (1, 2, 0)).astype(int) - r8.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_saliency_other_aspect_ratio(engine, synthetic_sd):
    fr = synth.blob_frames(2, 187, 250, seed=4)           # 4:3 -> network 288x384
    assert U.get_optimal_out_size((187, 250)) == (288, 384)
    maps = engine.saliency(torch.from_numpy(fr).cuda()).cpu().numpy()
    ref = U.saliency_u8(synthetic_sd, fr)
    d = np.abs(np.transpose(maps, (1, 2, 0)).astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_fused_threshold_entry_gives_the_bytes_of_the_two_calls(engine):
    fr = torch.from_numpy(synth.blob_frames(5, 140, 250, seed=12)).cuda()
    for t in (1, 90, 120, 255):
        two = engine.threshold_(engine.saliency(fr), t)
        assert torch.equal(engine.saliency(fr, threshold=t), two)
    assert torch.equal(engine.saliency(fr, threshold=0), engine.saliency(fr))


@pytest.mark.parametrize('shape', [(140, 250), (187, 250), (250, 140)])
def test_saliency_batch_and_chunk_independence(engine, shape):
    """A frame's map does not depend on its batch or its place in it -- also where a level's pixel count is not a multiple of
    32, so that the 32-pixel workgroups of k_pwpw / k_pw_sk straddle frames (the 4:3 and portrait geometries)."""
    h, w = shape
    fr = torch.from_numpy(synth.blob_frames(40, h, w, seed=9)).cuda()           # 40 > default chunk of 32
    full = engine.saliency(fr)
    assert torch.equal(engine.saliency(fr[3:4])[0], full[3])
    assert torch.equal(engine.saliency(fr[33:40]), full[33:40])
    assert torch.equal(engine.saliency(fr[5:18]), full[5:18])


@pytest.mark.parametrize('knobs', [
    {'SVC_FUSE_MAX': '0'},                                  # no fused inverted-residual blocks: k_pw / k_dw* / k_pw only
    {'SVC_DWPW': '0'},                                      # depthwise and project as two kernels
    {'SVC_DWPW': '0', 'SVC_DW_TILE': '0'},                  # ... with the one-output-per-thread depthwise
    {'SVC_DWPW_MIN_PX': '1'},                               # fused depthwise+project on the 8x13 level too
    {'SVC_PWR': '0'},                                       # short-K layers through k_pw (operands from global memory) instead of k_pwr
    {'SVC_PWR_NT': '1'},                                    # k_pwr with one column tile per workgroup
    {'SVC_PWR_NT': '4', 'SVC_PW_SK': '0'},                  # ... with four; long-K layers without split-K
    {'SVC_PWR': '0', 'SVC_PW_SK': '0', 'SVC_PW16': '0'},    # no k_pwr, no split-K, no 16x16x4 pointwise form
    {'SVC_PW_SMALL': '1'},                                  # 16x16x4 wave tiles for the small-M levels (three shapes)
    {'SVC_PW_SMALL': '2'},
    {'SVC_PW_SMALL': '3', 'SVC_PW_TR': '1'},               # ... and the float4 epilogue for every k_pw launch
    {'SVC_PW_TR': '0'},                                     # pointwise kernel with a lane per channel (scalar epilogue)
    {'SVC_FUSE_MAX': '13'},                                 # every block that can be fused is
    {'SVC_SPLIT_UP': '0'},                                  # decoder: up-sample + concatenate + one GEMM (the reference's order)
    {'SVC_IRB_FIXED': '0'},                                 # generic (run-time shaped) fused block instead of the fixed-shape instances
    {'SVC_SMOOTH_MFMA': '0'},                               # the 41x41 smoothing as the FMA kernel (the fall-back for geometries whose maps are not 8 x the low-resolution grid)
    {'SVC_FRONT': '0'},                                     # LANCZOS, features.0 and features.1 as three kernels instead of k_front
    {'SVC_CHUNK': '5'},                                     # ragged chunks of the batch
    {'SVC_SK_LANE': '0'},                                   # split-K layers read the [N][K] weight matrix instead of its lane-order copy
    {'SVC_PWPW': '0'},                                      # the skip branches' two 1x1 convolutions as two launches instead of k_pwpw
])
def test_saliency_kernel_families_agree(engine, synthetic_sd, knobs):
    """Every kernel family that can serve a layer (selected by shape at run time, forced here through
    the tuning knobs) must produce the same network output within the fp32 tolerance."""
    fr = torch.from_numpy(synth.blob_frames(7, 140, 250, seed=21)).cuda()
    ref_maps = engine.saliency(fr).cpu().numpy()
    ref_dec = engine.tap(ops.TAP_DEC, 3, (32, 52, 64))
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        other = ops.Engine(synthetic_sd)                     # the knobs are read when the handle is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        maps = other.saliency(fr).cpu().numpy()
        # taps address a frame of the last chunk, so they are only comparable when the batch is one chunk
        dec = other.tap(ops.TAP_DEC, 3, (32, 52, 64)) if 'SVC_CHUNK' not in knobs else None
    finally:
        other.close()
    if dec is not None:
        assert np.abs(dec - ref_dec).max() <= 2e-4 * np.abs(ref_dec).max()
    d = np.abs(maps.astype(int) - ref_maps.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


@pytest.mark.parametrize('shape', [(140, 250), (187, 250), (250, 140)])
def test_lane_order_weight_copies_are_bit_identical(engine, synthetic_sd, shape):
    """Round 4 (default on): k_pw_sk, k_dwpw, k_pwpw and k_front read their weights from lane-order copies of the matrices (a wave's
    load = 8 whole cache lines instead of 32 quarter-used ones).  Same values in the same order: the maps and the decoder tap equal
    those of the [N][K] reads (SVC_SK_LANE=0) bit for bit, at the three geometries."""
    h, w = shape
    NH, NW = U.get_optimal_out_size((h, w))
    fr = torch.from_numpy(synth.blob_frames(5, h, w, seed=3 * h + w)).cuda()
    maps = engine.saliency(fr).cpu().numpy()
    tap = engine.tap(ops.TAP_DEC, 4, (NH // 8, NW // 8, 64))
    old = os.environ.get('SVC_SK_LANE')
    os.environ['SVC_SK_LANE'] = '0'
    try:
        other = ops.Engine(synthetic_sd)
    finally:
        if old is None:
            os.environ.pop('SVC_SK_LANE', None)
        else:
            os.environ['SVC_SK_LANE'] = old
    try:
        assert np.array_equal(other.saliency(fr).cpu().numpy(), maps)
        assert np.array_equal(other.tap(ops.TAP_DEC, 4, (NH // 8, NW // 8, 64)), tap)
    finally:
        other.close()


@pytest.mark.parametrize('shape', [(140, 250), (187, 250), (250, 140), (360, 640), (97, 131)])
def test_front_kernel_bit_identical_to_three_kernels(engine, synthetic_sd, shape):
    """k_front (LANCZOS + features.0 + features.1 in one kernel) keeps the operation order of the three kernels it
    replaces: network input, feature taps and maps are bit-identical, at up-scaling, down-scaling and odd source sizes."""
    h, w = shape
    NH, NW = U.get_optimal_out_size((h, w))
    fr = torch.from_numpy(np.random.RandomState(h * w).randint(0, 256, (5, h, w, 3)).astype(np.uint8)).cuda()
    maps = engine.saliency(fr).cpu().numpy()
    taps = [engine.tap(ops.TAP_INPUT, 4, (NH, NW, 3)), engine.tap(ops.TAP_FEAT4X, 4, (NH // 8, NW // 8, 64)),
            engine.tap(ops.TAP_DEC, 4, (NH // 8, NW // 8, 64))]
    assert engine.front_fused() == (h <= NH and w <= NW)      # down-scaling geometries run the three kernels
    old = os.environ.get('SVC_FRONT')
    os.environ['SVC_FRONT'] = '0'
    try:
        other = ops.Engine(synthetic_sd)
    finally:
        if old is None:
            os.environ.pop('SVC_FRONT', None)
        else:
            os.environ['SVC_FRONT'] = old
    try:
        maps0 = other.saliency(fr).cpu().numpy()
        taps0 = [other.tap(ops.TAP_INPUT, 4, (NH, NW, 3)), other.tap(ops.TAP_FEAT4X, 4, (NH // 8, NW // 8, 64)),
                 other.tap(ops.TAP_DEC, 4, (NH // 8, NW // 8, 64))]
        assert not other.front_fused()
    finally:
        other.close()
    for a, b in zip(taps, taps0):
        assert np.array_equal(a, b)
    assert np.array_equal(maps, maps0)


def test_blend_chain_carried_over_between_calls(engine, golden_dir):
    """SVC_MAP_HELD: a blend chain that straddles two calls (its last map left for the second call, the already final map
    before it carried over as HELD | BLEND_NEXT) gives the same maps and centres as the chain in one call."""
    CP = P.init_crop_params()
    g = np.load(os.path.join(golden_dir, 'unisal_golden.npz'))
    base = np.transpose(g['smaps_u8'], (2, 0, 1)).copy()
    rng = np.random.RandomState(11)
    maps = np.stack([np.roll(base[i % len(base)], (3 * i, 5 * i), (0, 1)) for i in range(6)])
    maps[4] = (rng.rand(140, 250) < 0.02) * 200
    one = torch.from_numpy(maps.copy()).cuda()
    engine.threshold_(one, CP['t_threshold'])
    thr = one.clone()
    xy1 = engine.cluster_center_(one, [1, 1, 0, 0, 1, 0], CP).cpu().numpy()            # chains 0 -> 1 -> 2 and 4 -> 5
    H, B = ops.MAP_HELD, ops.BLEND_NEXT
    a = thr.clone()
    xya = engine.cluster_center_(a, [B, 0, H, 0, B, H], CP).cpu().numpy()              # maps 2 and 5 left for later
    assert torch.equal(a[[0, 1, 3, 4]], one[[0, 1, 3, 4]]) and torch.equal(a[[2, 5]], thr[[2, 5]])
    for i in (0, 1, 3, 4):
        assert np.array_equal(xya[i], xy1[i], equal_nan=True)
    b = torch.stack([a[1], a[2], a[4], a[5]])                                          # (final, raw) pairs of the two chains
    keep = b.clone()
    xyb = engine.cluster_center_(b, [H | B, 0, H | B, 0], CP).cpu().numpy()
    assert torch.equal(b[0], keep[0]) and torch.equal(b[2], keep[2])                   # held maps are not touched
    assert torch.equal(b[1], one[2]) and torch.equal(b[3], one[5])
    assert np.array_equal(xyb[1], xy1[2], equal_nan=True) and np.array_equal(xyb[3], xy1[5], equal_nan=True)
    c = keep.clone()                                                                    # a call with held maps only is a no-op
    engine.cluster_center_(c, [H, H, H, H], CP)
    assert torch.equal(c, keep)


def test_tail_bit_exact_default_settings(engine, golden_dir):
    CP = P.init_crop_params()
    g = np.load(os.path.join(golden_dir, 'unisal_golden.npz'))
    base = np.transpose(g['smaps_u8'], (2, 0, 1)).copy()
    rng = np.random.RandomState(3)
    extra = [np.zeros((140, 250), np.uint8)]                                          # empty -> None
    m = np.zeros((140, 250), np.uint8); m[10:13, 10:14] = 200; extra.append(m)        # 12 points: unclustered, unclosed
    m = np.zeros((140, 250), np.uint8); m[3:6, 3:12] = 200; extra.append(m)           # exactly hdbscan_min + 1
    m = np.zeros((140, 250), np.uint8); m[3:6, 3:12] = 200; m[8, 3] = 130; extra.append(m)   # hdbscan_min + 2
    m = (rng.rand(140, 250) < 0.03) * rng.randint(120, 256, (140, 250)); m[50:80, 100:160] = 250
    extra.append(m.astype(np.uint8))                                                  # speckle + block
    m = base[1].copy(); m[m < 60] = 0; extra.append(m)                                # large regions
    extra.append(np.full((140, 250), 200, np.uint8))                                  # every pixel set: N = 35000
    maps = np.concatenate([base, np.stack(extra)])
    flags = np.zeros(len(maps), np.uint8)
    flags[[0, 1, 5, 8]] = 1                                                           # blend chains 0->1->2, 5->6, 8->9
    stats = _check_tail(engine, maps, flags, CP)
    assert stats[4, 0] == 0 and stats[5, 0] == 12 and stats[-1, 0] > 12 * 1024
    # intermediate state of one map: core distances and the MST in Prim order
    st = engine.cluster_state(3, 35000)
    X = np.stack([st['pts'] & 255, (st['pts'] >> 8) & 255], 1).astype(np.int64)
    core = H.core_distances(X, H.effective_min_samples(len(X), 26, None))
    assert np.array_equal(core, st['core'])
    assert np.array_equal(np.stack(H.prim_mst(X, core), 1), st['mst'])


def _prim_and_labels(engine, maps, CP):
    """Per map: the device's Prim edge list and labels against the oracle's (the rare paths of k_prim_lvl / k_tree_par)."""
    dm = torch.from_numpy(maps.copy()).cuda()
    engine.cluster_center_(dm, None, CP)
    mcs, ms = CP['hdbscan_min'], CP['hdbscan_min_samples']
    for i in range(maps.shape[0]):
        st = engine.cluster_state(i, maps.shape[1] * maps.shape[2])
        X = np.stack([st['pts'] & 255, (st['pts'] >> 8) & 255], 1).astype(np.int64)
        assert np.array_equal(X, np.argwhere(maps[i] > 0))
        if len(X) <= mcs + 1:
            continue
        core = H.core_distances(X, H.effective_min_samples(len(X), mcs, ms))
        assert np.array_equal(core, st['core']), i
        u, v, w = H.prim_mst(X, core)
        assert np.array_equal(np.stack([u, v, w], 1), st['mst']), 'Prim sequence of map %d (N = %d)' % (i, len(X))
        ref = H.hdbscan_labels(X, mcs, ms)
        got = st['labels']
        assert np.array_equal(ref < 0, got < 0), i
        pairs = {(a, b) for a, b in zip(ref[ref >= 0], got[got >= 0])}
        assert len(pairs) == len({a for a, _ in pairs}) == len({b for _, b in pairs}), 'labels of map %d' % i


def test_prim_rounds_and_parallel_hierarchy_on_adversarial_maps(engine):
    """k_prim_lvl (rounds of up to 64 nodes, drops, rises, jumps beyond the ring table, the batch table running full)
    and k_tree_par (nearest greater ranks, pointer jumping) against the oracle's one-node-per-step Prim and its
    union-find hierarchy: sparse noise (every step a jump), dense noise, lines, lattices, blobs with outliers."""
    rng = np.random.RandomState(11)
    maps = []
    for dens in (0.002, 0.01, 0.03, 0.08, 0.2):
        maps.append(((rng.rand(140, 250) < dens) * 200).astype(np.uint8))
    m = np.zeros((140, 250), np.uint8); m[70, :] = 200; m[:, 125] = 180; maps.append(m)             # a cross of lines
    m = np.zeros((140, 250), np.uint8); m[::3, ::3] = 200; maps.append(m)                            # a lattice (d2 = 9 everywhere)
    m = np.zeros((140, 250), np.uint8); m[::7, ::5] = 200; m[40:60, 100:140] = 220; maps.append(m)   # sparse lattice + block
    ys, xs = np.mgrid[0:140, 0:250]
    m = np.zeros((140, 250), np.uint8)
    for (cy, cx, r) in ((30, 40, 12), (100, 200, 18), (70, 120, 9)):
        m[(ys - cy) ** 2 + (xs - cx) ** 2 < r * r] = 210
    m[rng.rand(140, 250) < 0.003] = 150; maps.append(m)                                              # three blobs + outliers
    maps = np.stack(maps)
    _prim_and_labels(engine, maps, P.init_crop_params())
    _prim_and_labels(engine, maps, dict(P.init_crop_params(), hdbscan_min=5, hdbscan_min_samples=3))
    small = np.stack([((rng.rand(35, 62) < d) * 200).astype(np.uint8) for d in (0.05, 0.2, 0.5, 0.9)])
    _prim_and_labels(engine, small, dict(P.init_crop_params(), hdbscan_min=5, hdbscan_min_samples=3))
    _prim_and_labels(engine, small, dict(P.init_crop_params(), hdbscan_min=2, hdbscan_min_samples=1))


def test_tail_bit_exact_other_parameters(engine):
    rng = np.random.RandomState(5)
    maps = []
    for s in range(6):
        m = np.zeros((140, 250), np.uint8)
        for _ in range(rng.randint(1, 5)):
            cy, cx, ry, rx = rng.randint(15, 125), rng.randint(20, 230), rng.randint(4, 14), rng.randint(4, 20)
            ys, xs = np.mgrid[0:140, 0:250]
            blob = (((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2) < 1
            m[blob] = rng.randint(90, 256)
        m[rng.rand(140, 250) < 0.004] = rng.randint(90, 256)
        maps.append(m)
    maps = np.stack(maps)
    best = dict(P.init_crop_params(), t_threshold=90, hdbscan_min=5, hdbscan_min_samples=3, select_sum=1)
    _check_tail(engine, maps, None, best)
    # the published best-settings set: cluster on maps shrunk by 4 (INTER_LINEAR down and up), nearest-shrunk centre
    _check_tail(engine, maps, np.array([1, 1, 0, 0, 1, 0], np.uint8), P.init_crop_params(True))
    _check_tail(engine, maps, None, dict(P.init_crop_params(), resize_factor=2))
    _check_tail(engine, maps, None, dict(P.init_crop_params(True), clust_filt=False))
    _check_tail(engine, maps, np.array([1, 0, 1, 1, 0, 0], np.uint8), dict(P.init_crop_params(), op_close=False))
    _check_tail(engine, maps, None, dict(P.init_crop_params(), clust_filt=False))
    _check_tail(engine, maps, None, dict(P.init_crop_params(), hdbscan_min=40, hdbscan_min_samples=10))


def test_tail_ragged_sizes_and_sparse_points(engine):
    rng = np.random.RandomState(8)
    CP = P.init_crop_params()
    for (h, w) in [(187, 250), (250, 140), (33, 47)]:
        m = np.zeros((3, h, w), np.uint8)
        m[0, h // 4:h // 2, w // 4:w // 2] = 200
        m[1][rng.rand(h, w) < 0.02] = 180              # sparse: core distances beyond the ring table
        m[2, :, :] = (rng.rand(h, w) < 0.5) * 150
        _check_tail(engine, m, np.array([0, 1, 0], np.uint8), CP)


def test_tail_randomised_maps_bit_exact(engine):
    """Many small maps of mixed texture (blobs, speckle, stripes, ties everywhere) and several parameter sets:
    point lists, core distances, Prim order, hierarchy, selection, CLOSE and centres must all agree with the oracle."""
    rng = np.random.RandomState(2024)
    for trial, (mcs, ms, ssum, close) in enumerate([(26, None, 2, True), (5, 3, 1, True), (12, 4, 2, False), (3, 2, 1, True)]):
        h, w = int(rng.randint(24, 70)), int(rng.randint(24, 90))
        maps = np.zeros((8, h, w), np.uint8)
        ys, xs = np.mgrid[0:h, 0:w]
        for i in range(8):
            m = rng.rand(h, w) < rng.choice([0.0, 0.03, 0.15, 0.5])
            for _ in range(rng.randint(0, 4)):
                cy, cx, ry, rx = rng.randint(0, h), rng.randint(0, w), rng.randint(2, 12), rng.randint(2, 16)
                m |= (((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2) < 1
            if i == 5:
                m = (xs % 4 == 0)                                  # stripes: every distance ties
            maps[i] = np.where(m, rng.randint(121, 256, (h, w)), rng.randint(0, 120, (h, w))).astype(np.uint8)
        flags = (rng.rand(8) < 0.4).astype(np.uint8)
        CP = dict(P.init_crop_params(), hdbscan_min=mcs, hdbscan_min_samples=ms, select_sum=ssum, op_close=close)
        _check_tail(engine, maps, flags, CP)


def test_tail_maximum_size_all_pixels_set(engine):
    """N = 35 000 points (every pixel above the threshold): the largest problem a 140x250 map can pose.
    Takes the global-memory forms of Prim (N > 32 768) and of the hierarchy (N > 4 352), and every
    mutual-reachability distance ties with its neighbours', so the Prim / sort tie rules decide the tree."""
    CP = P.init_crop_params()
    m = np.full((1, 140, 250), 200, np.uint8)
    m[0, 40:60, 100:130] = 0                           # a hole and a shorter first row: N = 34 390
    m[0, 0, 0:10] = 0
    _check_tail(engine, m, None, CP)                   # (the O(N^2) oracle needs about a minute for this one map)


def _dense_maps(rng, n, lo, hi):
    """140 x 250 maps with between lo and hi points: big soft blobs, dense noise, noise inside a blob, stripes with holes."""
    ys, xs = np.mgrid[0:140, 0:250]
    out = []
    while len(out) < n:
        kind = len(out) % 4
        m = np.zeros((140, 250), np.float32)
        if kind in (0, 2):
            for _ in range(rng.randint(1, 4)):
                cy, cx = rng.uniform(20, 120), rng.uniform(30, 220)
                ry, rx = rng.uniform(25, 90), rng.uniform(40, 160)
                m = np.maximum(m, 255 * np.exp(-(((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2) * rng.uniform(0.6, 2.0)))
            if kind == 2:
                m = m * (rng.rand(140, 250) < rng.uniform(0.5, 0.9))          # a blob with holes: core distances vary
        elif kind == 1:
            m = 255.0 * (rng.rand(140, 250) < rng.uniform(0.25, 0.95))        # dense noise
        else:
            m = 255.0 * (np.sin(xs / rng.uniform(3, 11)) * np.sin(ys / rng.uniform(3, 9)) > rng.uniform(-0.6, 0.2))
        u = np.clip(m + rng.uniform(0, 30) * rng.rand(140, 250), 0, 255).astype(np.uint8)
        u[u < 120] = 0
        if lo <= int((u > 0).sum()) <= hi:
            out.append(u)
    return np.stack(out)


def test_maps_beyond_8192_points_new_kernels_equal_the_round2_kernels_and_the_oracle(engine):
    """Round 4: maps of more than 8 192 points run the level-bucketed Prim with its per-point state in the workspace
    (k_prim_lvl_big) and the path-structured hierarchy with its per-edge arrays there (tp_body<2>) instead of the
    one-node-per-step Prim and the serial union-find.  24 maps of 8.5 - 22 k points (blobs, dense noise, blobs with holes,
    lattices) x both parameter sets against those round-2 kernels (Prim edge list, labels, maps, centres: identical), one
    ~9 k-point map against the oracle, and no map left to the fall-back kernels (hdr[23] = done by k_tree_par)."""
    import os
    saved = {k: os.environ.get(k) for k in ('SVC_PRIM_LVL', 'SVC_TREE_PAR', 'SVC_TAIL_MERGE')}
    try:
        os.environ.update(SVC_PRIM_LVL='0', SVC_TREE_PAR='0', SVC_TAIL_MERGE='0')
        old = ops.Engine(seed=0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        rng = np.random.RandomState(77)
        maps = _dense_maps(rng, 24, 8500, 22000)
        flags = np.zeros(24, np.uint8)
        flags[[3, 4, 11]] = 1
        for CP in (P.init_crop_params(), dict(P.init_crop_params(), hdbscan_min=5, hdbscan_min_samples=3, select_sum=1)):
            a, b = torch.from_numpy(maps).cuda(), torch.from_numpy(maps).cuda()
            xa, sa = old.cluster_center_(a, flags, CP, want_stats=True)
            xb, sb = engine.cluster_center_(b, flags, CP, want_stats=True)
            assert torch.equal(a, b) and torch.equal(sa, sb)
            assert np.array_equal(xa.cpu().numpy(), xb.cpu().numpy(), equal_nan=True)
            assert int(sb[:, 0].min()) > 8192
            for i in range(0, 24, 3):
                s_old, s_new = old.cluster_state(i, 35000), engine.cluster_state(i, 35000)
                assert np.array_equal(s_old['mst'], s_new['mst']), 'Prim sequence of map %d (N = %d)' % (i, s_old['n'])
                assert np.array_equal(s_old['labels'], s_new['labels']), 'labels of map %d' % i
                assert s_new['hdr'][16] > 0                                    # Prim in rounds (k_prim_lvl_big)
                if CP['hdbscan_min'] == 26:                                    # (min_cluster_size 5 on 20 k points: more clusters than the
                    assert s_new['hdr'][23] == 1                               #  LDS tables hold -> the serial builder; not a shipped setting)
        small = _dense_maps(np.random.RandomState(78), 1, 8300, 9500)
        _check_tail(engine, small, None, P.init_crop_params())                  # (the O(N^2) oracle: ~10 s)
    finally:
        old.close()


def test_largest_geometries_up_to_65025_points_equal_the_round2_kernels(engine):
    """What a 140 x 250 map cannot reach: more new tree nodes than k_prim_lvl_big's batch table holds (> 36 864 since the last
    rise), the fourth level of the hierarchy's nearest-greater search (> 8 192 x 4 edges), a 255 x 255 map with every pixel set
    (65 025 points).  Against the round-2 kernels: Prim edge list, labels, maps, centres identical (tools/soak_big_maps.py is
    the long form: four geometries x 8 maps x two parameter sets)."""
    import os
    saved = {k: os.environ.get(k) for k in ('SVC_PRIM_LVL', 'SVC_TREE_PAR', 'SVC_TAIL_MERGE')}
    try:
        os.environ.update(SVC_PRIM_LVL='0', SVC_TREE_PAR='0', SVC_TAIL_MERGE='0')
        old = ops.Engine(seed=0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        rng = np.random.RandomState(91)
        for (h, w) in ((187, 250), (255, 255)):
            maps = np.zeros((3, h, w), np.uint8)
            maps[0] = 200                                                    # every pixel: one plateau longer than the batch table
            maps[1] = np.where(rng.rand(h, w) < 0.8, 220, 0)
            ys, xs = np.mgrid[0:h, 0:w]
            maps[2] = np.where(np.sin(xs / 4.0) * np.sin(ys / 3.0) > -0.7, 180, 0)
            CP = P.init_crop_params()
            a, b = torch.from_numpy(maps).cuda(), torch.from_numpy(maps).cuda()
            xa, sa = old.cluster_center_(a, None, CP, want_stats=True)
            xb, sb = engine.cluster_center_(b, None, CP, want_stats=True)
            assert torch.equal(a, b) and torch.equal(sa, sb) and int(sb[:, 0].min()) > 30000
            assert np.array_equal(xa.cpu().numpy(), xb.cpu().numpy(), equal_nan=True)
            for i in range(3):
                s_old, s_new = old.cluster_state(i, h * w), engine.cluster_state(i, h * w)
                assert np.array_equal(s_old['mst'], s_new['mst']), (h, w, i, s_old['n'])
                assert np.array_equal(s_old['labels'], s_new['labels']), (h, w, i)
    finally:
        old.close()


def test_tail_batch_independence_at_full_size(engine):
    """Size-independent property at BASELINE config 2 (B=32, 640x360): every map's result is
    independent of the batch it is processed in, and filtering only ever removes/closes."""
    CP = P.init_crop_params()
    fr = torch.from_numpy(synth.blob_frames(32, 360, 640, seed=0)).cuda()
    maps = engine.saliency(engine.resize_frames(fr, 140, 250))
    engine.threshold_(maps, CP['t_threshold'])
    thr = maps.clone()
    xy = engine.cluster_center_(maps, None, CP)
    for i in (0, 7, 31):
        one = thr[i:i + 1].clone()
        xy1 = engine.cluster_center_(one, None, CP)
        assert torch.equal(one[0], maps[i]) and torch.equal(xy1[0], xy[i])
    t = thr.cpu().numpy()
    closed = np.stack([cv_ref.morph_close_5x5(x) for x in t])
    assert (maps.cpu().numpy() <= closed).all()
    xyc = xy.cpu().numpy()
    assert (xyc[:, 0] >= 0).all() and (xyc[:, 0] <= 249).all() and (xyc[:, 1] >= 0).all() and (xyc[:, 1] <= 139).all()


def test_c_abi_error_and_empty_conventions(engine):
    """Status code + svc_last_error() instead of exceptions / exit (include/svc.h); n = 0 is a no-op."""
    from retargetvid_amd._lib import SvcError
    CP = P.init_crop_params()
    empty = torch.empty((0, 140, 250, 3), dtype=torch.uint8, device='cuda')
    assert engine.saliency(empty).shape == (0, 140, 250)
    assert engine.resize_frames(torch.empty((0, 360, 640, 3), dtype=torch.uint8, device='cuda'), 140, 250).shape[0] == 0
    assert engine.cluster_center_(torch.empty((0, 140, 250), dtype=torch.uint8, device='cuda'), None, CP).shape == (0, 2)
    big = torch.zeros((1, 300, 300), dtype=torch.uint8, device='cuda')
    with pytest.raises(SvcError, match='exceeds the supported'):
        engine.cluster_center_(big, None, CP)
    m = torch.zeros((1, 140, 250), dtype=torch.uint8, device='cuda')
    with pytest.raises(SvcError, match='hdbscan_min'):
        engine.cluster_center_(m, None, dict(CP, hdbscan_min=1))
    with pytest.raises(TypeError):
        engine.saliency(torch.zeros((1, 140, 250, 3), dtype=torch.uint8))         # host tensor: the path is device-only
    xy = engine.cluster_center_(m, None, CP).cpu().numpy()                          # an all-zero map has no centre
    assert np.isnan(xy).all()
    assert engine.saliency(torch.zeros((1, 140, 250, 3), dtype=torch.uint8, device='cuda')).shape == (1, 140, 250)


def test_iou_bit_exact(engine):
    rng = np.random.RandomState(0)
    a = rng.randint(-5, 600, (5000, 4)).astype(np.int32); a[:, 2:] += np.abs(a[:, :2])
    b = rng.randint(-5, 600, (5000, 4)).astype(np.int32); b[:, 2:] += np.abs(b[:, :2])
    got = ops.iou_boxes(a, b)
    ref = np.array([T.iou(x, y) for x, y in zip(a.tolist(), b.tolist())])
    assert np.array_equal(got, ref)
    from retargetvid_amd import smartVidCrop as S
    assert S.bb_intersection_over_union([0, 0, 9, 9], [5, 0, 14, 9]) == T.iou([0, 0, 9, 9], [5, 0, 14, 9])


def test_device_edge_order_equals_numpy_argsort(engine, golden_dir):
    """k_sort's routine (numpy's default argsort -- an unstable introsort -- emulated in parallel) against numpy's own
    permutations: tie-heavy, sorted, reversed, random and adversarial arrays (the latter reach the depth limit, i.e. the
    heapsort fall-back), the LDS and the global-memory form (n > ~6000), bit for bit."""
    from oracle import npsort_ref
    g = np.load(os.path.join(golden_dir, 'npsort_golden.npz'))
    for i in range(int(g['n'])):
        w, o = g['w_%d' % i], g['o_%d' % i]
        assert np.array_equal(engine.argsort_u32(w.astype(np.uint32)), o), (i, len(w))
    rng = np.random.RandomState(11)
    for n, hi in ((6500, 12), (20000, 40), (34999, 7), (34999, 100000)):
        w = rng.randint(1, hi, n).astype(np.uint32)
        assert np.array_equal(engine.argsort_u32(w), np.array(npsort_ref.argsort(w.tolist()))), (n, hi)


# ---- every network layer, every frame, three geometries, checkpoints without the luminance carrier -----------------
_TAPS = (('feat_4x', 'TAP_FEAT4X', 8, 64), ('feat_2x', 'TAP_FEAT2X', 16, 160), ('feat_1x', 'TAP_FEAT1X', 32, 1296),
         ('post_cnn', 'TAP_POSTCNN', 32, 256), ('dec', 'TAP_DEC', 8, 64))
# fp32 against fp32 in another summation order: the error is a noise floor proportional to the tensor's scale, not to
# the element (sums of hundreds of +-O(1) terms cancel), so the elementwise bound is atol + rtol |ref| with atol a
# fraction of max|ref|; the mean error is bounded separately, an order of magnitude lower.  Measured on MI355X
# (tools/net_error_report.py): synthetic checkpoints max 3.2e-5 / mean 3.0e-6 of max|ref|, the reference-initialised one
# (unit-variance activations after calibrated BatchNorm, deeper cancellation) max 1.6e-4 / mean 2.4e-5; u8 maps differ
# by one grey level on 0.03 % / 0.4 % of the pixels.
_TOL = {'nc': (8e-5, 8e-6, 1e-3), 'ri': (4e-4, 6e-5, 1e-2), 'tl': (4e-4, 6e-5, 2e-3), 'tl2': (4e-4, 6e-5, 2e-3)}


@pytest.mark.parametrize('ck,pipe', [('nc', 'bf16x6'), ('ri', 'bf16x6'), ('tl', 'bf16x6'), ('tl2', 'bf16x6'), ('ri', 'f32'), ('tl', 'f32')])
def test_network_every_layer_every_frame_three_geometries(ck, pipe, golden_dir):
    """Every tap of every frame at three geometries against the oracle AND the reference model's own goldens, per checkpoint family.
    pipe = 'bf16x6' (the default since round 5): the 1x1 convolutions on the bf16 matrix pipe with split operands (csrc/svc_net.hip
    "Split-bf16 operands"), tolerances as they were for the fp32 pipe; pipe = 'f32' (SVC_MX=f32, rounds 1-4): the same gates on the two
    checkpoints with realistic activations."""
    from retargetvid_amd import weights
    g = np.load(os.path.join(golden_dir, {'tl': 'unisal_golden3.npz', 'tl2': 'unisal_golden4.npz'}.get(ck, 'unisal_golden2.npz')))
    if ck in ('tl', 'tl2'):                            # trained-like: the reference model fitted to blob targets (peaky maps); two fits
        sd = weights.make_trained_like_state_dict(golden_dir, variant=1 if ck == 'tl' else 2)
    elif ck == 'nc':
        sd = weights.make_synthetic_state_dict(3, carrier=False)
    else:
        sd = weights.make_reference_init_state_dict(7, {k[3:]: g[k] for k in g.files if k.startswith('bn/')})
    atol_f, mean_f, u8_frac = _TOL[ck]
    old_mx = os.environ.get('SVC_MX')
    os.environ['SVC_MX'] = pipe
    try:
        eng = ops.Engine(sd)
    finally:
        if old_mx is None:
            os.environ.pop('SVC_MX', None)
        else:
            os.environ['SVC_MX'] = old_mx
    assert eng.matrix_pipe() == pipe
    try:
        for gname in ('16x9', '4x3', 'port'):
            frames = g['frames_' + gname]
            h, w = frames.shape[1:3]
            NH, NW = U.get_optimal_out_size((h, w))
            maps = eng.saliency(torch.from_numpy(frames).cuda()).cpu().numpy()
            taps = {}
            ref_maps = U.saliency_u8(sd, frames, taps)
            for i in range(frames.shape[0]):
                t = taps['frames'][i]
                assert np.array_equal(eng.tap(ops.TAP_INPUT, i, (NH, NW, 3)), t['input'][0].permute(1, 2, 0).numpy())
                checks = [(key, eng.tap(getattr(ops, tap), i, (NH // div, NW // div, ch)), t[key][0].permute(1, 2, 0).numpy())
                          for key, tap, div, ch in _TAPS]
                checks.append(('pre', eng.tap(ops.TAP_PRE, i, (h, w)), t['pre'][0].numpy()))
                for key, got, ref in checks:
                    if key == 'feat_1x':
                        got = got[:, :, :1280]
                    scale = float(np.abs(ref).max())
                    d = np.abs(got - ref)
                    assert (d <= atol_f * scale + 1e-4 * np.abs(ref)).all(), (ck, gname, i, key, float(d.max() / scale))
                    assert d.mean() <= mean_f * scale, (ck, gname, i, key, float(d.mean() / scale))
                # u8 maps against the oracle and against the reference model's own output
                for r8 in (ref_maps[:, :, i], g['u8_%s_%s_%d' % (ck, gname, i)]):
                    du = np.abs(maps[i].astype(int) - r8.astype(int))
                    assert du.max() <= 1 and (du > 0).mean() < u8_frac, (ck, gname, i)
    finally:
        eng.close()


def _fuzz_maps(rng, n, h, w):
    """Random thresholded maps of many kinds: blobs of different sizes and counts, rings, stripes, noise at several
    densities, outliers, value plateaus and gradients."""
    ys, xs = np.mgrid[0:h, 0:w]
    out = np.zeros((n, h, w), np.uint8)
    for i in range(n):
        kind = i % 6
        m = np.zeros((h, w), np.float32)
        if kind in (0, 1, 2):
            for _ in range(rng.randint(1, 5)):
                cy, cx = rng.uniform(0, h), rng.uniform(0, w)
                ry, rx = rng.uniform(2, h / (3 if kind else 6)), rng.uniform(2, w / (4 if kind else 8))
                d = ((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2
                m = np.maximum(m, 255 * np.exp(-d * rng.uniform(0.5, 2.0)))
        elif kind == 3:
            cy, cx, r0 = rng.uniform(h * 0.3, h * 0.7), rng.uniform(w * 0.3, w * 0.7), rng.uniform(5, min(h, w) * 0.4)
            d = np.sqrt((ys - cy) ** 2 + (xs - cx) ** 2)
            m = 255 * np.exp(-((d - r0) / rng.uniform(1.0, 4.0)) ** 2)                       # a ring
        elif kind == 4:
            m = 255 * (np.sin(xs / rng.uniform(2, 9) + ys / rng.uniform(3, 15)) > rng.uniform(0.2, 0.95))   # stripes
        else:
            m = 255 * (rng.rand(h, w) < rng.choice([0.003, 0.02, 0.08, 0.3]))               # noise
        m = m + rng.uniform(0, 40) * rng.rand(h, w)
        m[rng.rand(h, w) < rng.choice([0.0, 0.001, 0.01])] = 255                              # outliers
        u = np.clip(m, 0, 255).astype(np.uint8)
        u[u < 120] = 0
        out[i] = u
    return out


def test_new_tail_kernels_equal_the_round2_kernels_on_random_maps():
    """k_prim_lvl + k_tree_par (round 3) against the one-node-per-step Prim and the serial union-find hierarchy of round 2
    (SVC_PRIM_LVL=0, SVC_TREE_PAR=0; both verified against the oracle by the tests above) on 288 random maps of four
    sizes and both parameter sets: Prim edge lists, labels, filtered maps and centres must be identical.  A wide net for
    the rare paths (drops, rises, jumps, full batch tables, maps above 4 352 points) that costs seconds on the device."""
    import os
    saved = {k: os.environ.get(k) for k in ('SVC_PRIM_LVL', 'SVC_TREE_PAR', 'SVC_TAIL_MERGE')}
    try:
        os.environ.update(SVC_PRIM_LVL='0', SVC_TREE_PAR='0', SVC_TAIL_MERGE='0')
        old = ops.Engine(seed=0)
        for k in saved:
            os.environ.pop(k, None)
        new = ops.Engine(seed=0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        rng = np.random.RandomState(31)
        n_maps = 0
        for (h, w, n) in ((140, 250, 48), (35, 62, 60), (187, 250, 24), (255, 255, 12)):    # 255 x 255: the largest map the tail takes
            maps = _fuzz_maps(rng, n, h, w)
            flags = (rng.rand(n) < 0.2).astype(np.uint8)
            flags[-1] = 0
            for CP in (P.init_crop_params(), dict(P.init_crop_params(), hdbscan_min=5, hdbscan_min_samples=3, select_sum=1)):
                a, b = torch.from_numpy(maps).cuda(), torch.from_numpy(maps).cuda()
                xa, sa = old.cluster_center_(a, flags, CP, want_stats=True)
                xb, sb = new.cluster_center_(b, flags, CP, want_stats=True)
                assert torch.equal(a, b), (h, w)
                assert torch.equal(sa, sb)
                assert np.array_equal(xa.cpu().numpy(), xb.cpu().numpy(), equal_nan=True)
                for i in range(0, n, 5):                                   # intermediate state of every fifth map
                    s_old, s_new = old.cluster_state(i, h * w), new.cluster_state(i, h * w)
                    assert np.array_equal(s_old['mst'], s_new['mst']), 'Prim sequence of map %d (%dx%d, N = %d)' % (i, h, w, s_old['n'])
                    assert np.array_equal(s_old['labels'], s_new['labels']), 'labels of map %d' % i
                n_maps += n
        assert n_maps == 288
    finally:
        old.close()
        new.close()


def test_the_fp32_pipe_stays_selectable_and_both_pipes_are_reproducible(engine, synthetic_sd, golden_dir):
    """SVC_MX=f32 selects the fp32 matrix pipe of rounds 1-4 (the default is bf16x6): the golden gates of
    test_saliency_against_oracle_and_reference_golden with tolerances unchanged, and bit-identical maps run after run and whatever the
    batch for BOTH pipes."""
    assert engine.matrix_pipe() == 'bf16x6'
    old = os.environ.get('SVC_MX')
    os.environ['SVC_MX'] = 'f32'
    try:
        f32 = ops.Engine(synthetic_sd)
    finally:
        if old is None:
            os.environ.pop('SVC_MX', None)
        else:
            os.environ['SVC_MX'] = old
    try:
        assert f32.matrix_pipe() == 'f32'
        g = np.load(os.path.join(golden_dir, 'unisal_golden.npz'))
        frames = g['frames']
        maps = f32.saliency(torch.from_numpy(frames).cuda()).cpu().numpy()
        taps = {}
        ref = U.saliency_u8(synthetic_sd, frames, taps)
        t = taps['frames'][0]
        for which, shape, key in [(ops.TAP_FEAT4X, (32, 52, 64), 'feat_4x'), (ops.TAP_FEAT2X, (16, 26, 160), 'feat_2x'),
                                  (ops.TAP_POSTCNN, (8, 13, 256), 'post_cnn'), (ops.TAP_DEC, (32, 52, 64), 'dec')]:
            r = t[key][0].permute(1, 2, 0).numpy()
            assert np.abs(f32.tap(which, 0, shape) - r).max() <= 2e-4 * np.abs(r).max(), key
        assert np.abs(f32.tap(ops.TAP_PRE, 0, (140, 250)) - t['pre'][0].numpy()).max() < 1e-4
        for r8 in (ref, g['smaps_u8']):
            d = np.abs(np.transpose(maps, (1, 2, 0)).astype(int) - r8.astype(int))
            assert d.max() <= 1 and (d > 0).mean() < 1e-3
        fr = torch.from_numpy(synth.blob_frames(40, 140, 250, seed=9)).cuda()
        for e in (f32, engine):
            full = e.saliency(fr)
            assert torch.equal(e.saliency(fr), full)
            assert torch.equal(e.saliency(fr[33:40]), full[33:40]) and torch.equal(e.saliency(fr[3:4])[0], full[3])
    finally:
        f32.close()


def test_maps_are_bit_reproducible_with_four_streams_sharing_the_chip(synthetic_sd):
    """Round 5: with the bf16 pipe the pre-softmax map had 16 pixels of one wavefront wrong in 1 - 6 % of the passes when four streams
    shared the chip -- the smoothing kernel's bilinear stage lost one product in a packed-instruction sequence beside workgroups of
    k_pwr's bf16 form; the stage runs on scalar instructions since (sd_bilinear), without the CU isolation that
    contained it first.  Four engines on four streams push the same 32 frames 120 times: every map equals the single-stream
    reference (tools/soak_network_concurrent.py is the long form: 0 of 20 400 passes without isolation; the packed form: 5 - 9 %)."""
    from retargetvid_amd import scheduler
    fr = torch.from_numpy(synth.blob_frames(32, 140, 250, seed=0)).cuda()
    engs = [ops.Engine(synthetic_sd) for _ in range(4)]
    try:
        sts = scheduler.lane_streams(torch.device('cuda', torch.cuda.current_device()), 4)
        ref = engs[0].saliency(fr).clone()
        outs = [torch.empty_like(ref) for _ in range(4)]
        for it in range(120):
            for i in range(4):
                with torch.cuda.stream(sts[i]):
                    engs[i].saliency(fr, out=outs[i])
            torch.cuda.synchronize()
            for i in range(4):
                assert torch.equal(outs[i], ref), (it, i)
    finally:
        for e in engs:
            e.close()
