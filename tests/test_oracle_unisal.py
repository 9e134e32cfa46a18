"""Pins oracle/unisal_ref.py and oracle/lanczos_ref.py to outputs of the REFERENCE model code
(tests/golden/unisal_golden.npz, made by tools/make_golden_unisal.py) and of Pillow."""
import os

import pytest

import numpy as np
import torch

from oracle import lanczos_ref, unisal_ref as U


def test_optimal_out_size():
    assert U.get_optimal_out_size((140, 250)) == (256, 416)      # 16:9 -> (8,13)*32, SURVEY fact 3
    assert U.get_optimal_out_size((360, 640)) == (256, 416)
    assert U.get_optimal_out_size((250, 250)) == (320, 320)
    assert U.get_optimal_out_size((250, 140)) == (416, 256)


def test_lanczos_matches_pillow_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'lanczos_golden.npz'))
    for i in range(4):
        a, ref = g['in_%d' % i], g['out_%d' % i]
        got = lanczos_ref.resize_lanczos_u8(a, ref.shape[0], ref.shape[1])
        assert np.array_equal(got, ref), 'case %d differs from Pillow %s' % (i, g['pillow_version'])


def test_lanczos_matches_installed_pillow():
    PIL = __import__('pytest').importorskip('PIL.Image')
    a = np.random.RandomState(1).randint(0, 256, (140, 250, 3)).astype(np.uint8)
    ref = np.asarray(PIL.fromarray(a).resize((416, 256), PIL.LANCZOS))
    assert np.array_equal(lanczos_ref.resize_lanczos_u8(a, 256, 416), ref)


def test_forward_matches_reference_model(golden_dir, synthetic_sd):
    torch.set_num_threads(1)             # the golden was generated single-threaded
    g = np.load(os.path.join(golden_dir, 'unisal_golden.npz'))
    frames = g['frames']
    taps = {}
    maps = U.saliency_u8(synthetic_sd, frames, taps)
    assert maps.shape == (140, 250, frames.shape[0]) and maps.dtype == np.uint8
    t0 = taps['frames'][0]
    assert np.array_equal(t0['input'][0].numpy(), g['input_0'])
    for k in ('feat_1x', 'feat_2x', 'feat_4x', 'post_cnn'):
        ref = g[k + '_0']
        assert np.abs(t0[k][0].numpy() - ref).max() <= 1e-4 * np.abs(ref).max(), k
    d = np.abs(maps.astype(int) - g['smaps_u8'].astype(int))
    # same torch ops as the reference: identical up to thread-count dependent summation order
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    for i in range(frames.shape[0]):
        lp = torch.log_softmax(taps['frames'][i]['pre'].reshape(1, -1), 1).reshape(140, 250).numpy()
        assert np.abs(lp - g['logp_%d' % i]).max() < 1e-4


def _golden2_checkpoint(g, ck):
    from retargetvid_amd import weights
    if ck == 'nc':
        return weights.make_synthetic_state_dict(3, carrier=False)
    stats = {k[3:]: g[k] for k in g.files if k.startswith('bn/')}
    return weights.make_reference_init_state_dict(7, stats)


def test_forward_matches_reference_model_without_carrier_all_geometries(golden_dir):
    """tests/golden/unisal_golden2.npz (tools/make_golden_unisal2.py): the reference model on a non-carrier random
    checkpoint and on a reference-initialised one (BatchNorm statistics calibrated by the reference model itself),
    at the 16:9, 4:3 and portrait network geometries, every frame: log-softmax maps, u8 maps, taps of frame 0."""
    torch.set_num_threads(4)
    g = np.load(os.path.join(golden_dir, 'unisal_golden2.npz'))
    for ck in ('nc', 'ri'):
        sd = _golden2_checkpoint(g, ck)
        for gname in ('16x9', '4x3', 'port'):
            frames = g['frames_' + gname]
            h, w = frames.shape[1:3]
            taps = {}
            maps = U.saliency_u8(sd, frames, taps)
            for i in range(frames.shape[0]):
                tag = '%s_%s_%d' % (ck, gname, i)
                t = taps['frames'][i]
                lp = torch.log_softmax(t['pre'].reshape(1, -1), 1).reshape(h, w).numpy()
                assert np.abs(lp - g['logp_' + tag]).max() < 2e-5, tag
                d = np.abs(maps[:, :, i].astype(int) - g['u8_' + tag].astype(int))
                assert d.max() <= 1 and (d > 0).mean() < 2e-3, tag
                if i == 0:
                    for k in ('feat_2x', 'post_cnn'):
                        ref = g['%s_%s' % (k, tag)]
                        assert np.allclose(t[k][0].numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max()), (k, tag)
                    ref = g['adapt_' + tag]
                    assert np.allclose(t['adapt'][0].numpy(), ref[0], rtol=1e-4, atol=1e-5 * np.abs(ref).max()), tag
    # the reference-initialised maps are not flat: the u8 map spans well over 100 grey levels
    assert np.ptp(g['u8_ri_16x9_0']) > 100


@pytest.mark.parametrize('variant', [1, 2])
def test_forward_matches_reference_model_on_the_trained_like_checkpoint(golden_dir, variant):
    """tests/golden/unisal_golden3.npz (tools/make_golden_unisal3.py): the reference model with its last decoder stage FITTED
    to blob targets -- peaky maps like a trained network's -- at the three geometries, every frame."""
    from retargetvid_amd import weights
    torch.set_num_threads(4)
    g = np.load(os.path.join(golden_dir, 'unisal_golden3.npz' if variant == 1 else 'unisal_golden4.npz'))
    sd = weights.make_trained_like_state_dict(golden_dir, variant=variant)
    for gname in ('16x9', '4x3', 'port'):
        frames = g['frames_' + gname]
        h, w = frames.shape[1:3]
        taps = {}
        maps = U.saliency_u8(sd, frames, taps)
        for i in range(frames.shape[0]):
            tag = '%s_%s_%d' % ('tl' if variant == 1 else 'tl2', gname, i)
            t = taps['frames'][i]
            lp = torch.log_softmax(t['pre'].reshape(1, -1), 1).reshape(h, w).numpy()
            assert np.abs(lp - g['logp_' + tag]).max() < 1e-4, tag               # the log-softmax spans ~40 here (a peaky map)
            d = np.abs(maps[:, :, i].astype(int) - g['u8_' + tag].astype(int))
            assert d.max() <= 1 and (d > 0).mean() < 2e-3, tag
            if i == 0:
                ref = g['adapt_' + tag]
                assert np.allclose(t['adapt'][0].numpy(), ref[0], rtol=1e-4, atol=1e-5 * np.abs(ref).max()), tag
    # peaky: a few hundred points above the default threshold, a handful of pixels per grey level next to it
    hist = g['level_hist_16x9'] / 24.0
    assert 100 < hist[120:].sum() < 3000 and hist[110:131].mean() < 20


def test_quantise_is_floor_of_scaled_softmax():
    x = torch.randn(2, 140, 250)
    q = U.quantise_u8(x)
    m = x.reshape(2, -1).max(1).values.reshape(2, 1, 1)
    approx = np.floor(255.0 * torch.exp(x - m).double().numpy())
    assert np.abs(q.astype(int) - approx).max() <= 1
    assert q.max() == 255
