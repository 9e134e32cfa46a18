"""TransNet V1 oracle (oracle/transnet_ref.py) and the host logic of retargetvid_amd/transnetv1_handler.py -- CPU only.
The oracle is PARITY UNPINNED against the reference's TensorFlow graph (no TensorFlow, no checkpoint here); these tests pin
its building blocks to independent restatements: the dilated SAME convolution against explicit loops, the windowing
against the property the reference's iterator has (every frame predicted once, from the middle of a window)."""
import numpy as np
import pytest

from oracle import transnet_ref as R
from retargetvid_amd import weights


def test_dilated_same_convolution_matches_explicit_loops():
    import torch
    import torch.nn.functional as F
    rng = np.random.RandomState(0)
    T, H, W, cin, cout = 11, 5, 6, 3, 4
    x = rng.randn(1, T, H, W, cin).astype(np.float32)
    for d in (1, 2, 4, 8):
        k = rng.randn(3, 3, 3, cin, cout).astype(np.float32)           # TensorFlow layout [kt, kh, kw, in, out]
        b = rng.randn(cout).astype(np.float32)
        ref = np.zeros((T, H, W, cout), np.float64)
        for t in range(T):
            for y in range(H):
                for xx in range(W):
                    acc = b.astype(np.float64).copy()
                    for kt in range(3):
                        for kh in range(3):
                            for kw in range(3):
                                tt, yy, x2 = t + (kt - 1) * d, y + kh - 1, xx + kw - 1
                                if 0 <= tt < T and 0 <= yy < H and 0 <= x2 < W:
                                    acc += x[0, tt, yy, x2].astype(np.float64) @ k[kt, kh, kw].astype(np.float64)
                    ref[t, y, xx] = np.maximum(acc, 0)
        xt = torch.from_numpy(x).permute(0, 4, 1, 2, 3)
        got = F.relu(F.conv3d(xt, torch.from_numpy(k).permute(4, 3, 0, 1, 2).contiguous(), torch.from_numpy(b),
                              padding=(d, 1, 1), dilation=(d, 1, 1)))[0].permute(1, 2, 3, 0).numpy()
        assert np.abs(got - ref).max() < 1e-4


@pytest.mark.parametrize('n', [1, 7, 49, 50, 51, 100, 130, 2037])
def test_windows_predict_every_frame_once_from_a_window_middle(n):
    from retargetvid_amd import transnetv1_handler as Hd
    wi = R.window_indices(n)
    assert np.array_equal(wi, Hd.window_indices(n))
    assert wi.shape[1] == 100 and len(wi) == -(-n // 50)
    kept = wi[:, 25:75].reshape(-1)[:n]
    assert np.array_equal(kept, np.arange(n))                       # the reference keeps [25:75] of every window, then [:n]
    assert (wi[0, :25] == 0).all() and (wi[-1, 75:] == n - 1).all()  # edge frames repeated


def test_synthetic_network_and_blob_layout():
    sd = weights.make_transnet_state_dict(3)
    cells = weights.transnet_cells()
    assert [(c[2], c[3]) for c in cells] == [(3, 16), (64, 16), (64, 32), (128, 32), (128, 64), (256, 64)]
    assert sd['TransNet/dense/kernel'].shape == (4608, 256) and sd['TransNet/dense_1/kernel'].shape == (256, 2)
    blob = weights.pack_transnet_blob(sd)
    want = sum(4 * ((f + 31) // 32 * 32) * ((27 * max(4, cin) + 7) // 8 * 8) + 4 * f for _, _, cin, f in cells) + 4608 * 256 + 256 + 512 + 2
    assert blob.dtype == np.float32 and blob.size == want
    # first cell, dilation 2: row = output channel, k = tap * 4 + channel; the fourth channel and the rows beyond 16 are zero
    k = sd['TransNet/SDDCNN_1/DDCNN_1/Conv3D_2/kernel']
    w = blob[32 * 112:2 * 32 * 112].reshape(32, 112)
    assert w[5, 13 * 4 + 2] == k[1, 1, 1, 2, 5] and (w[:, 3:108:4] == 0).all() and (w[16:] == 0).all() and (w[:, 108:] == 0).all()


def test_forward_shapes_and_probabilities():
    sd = weights.make_transnet_state_dict(0)
    fr = np.random.RandomState(1).randint(0, 256, (2, 20, 27, 48, 3)).astype(np.uint8)
    p = R.forward(sd, fr)
    assert p.shape == (2, 20) and p.dtype == np.float32 and (p > 0).all() and (p < 1).all()
    # a frame's prediction depends on the other frames of its window through the dilated temporal taps only
    fr2 = fr.copy(); fr2[0, 19] = 255 - fr2[0, 19]
    p2 = R.forward(sd, fr2)
    assert np.array_equal(p[1], p2[1]) and not np.allclose(p[0], p2[0])


def test_scene_walks():
    from retargetvid_amd import transnetv1_handler as Hd
    pred = np.zeros(60, np.float32); pred[[20, 21, 45]] = 0.9
    sc = Hd.scenes_from_predictions(pred, 0.1)
    assert sc.tolist() == [[0, 20], [22, 45], [46, 59]]
    assert np.array_equal(Hd.predictions_to_scenes(pred, 0.1), R.predictions_to_scenes(pred, 0.1))
    assert Hd.predictions_to_scenes(np.ones(9), 0.5).tolist() == [[0, 8]] == R.predictions_to_scenes(np.ones(9), 0.5).tolist()
    # assert_segmentation: shots shorter than 12 frames dropped, neighbours made adjacent, the last one reaches the end
    shots = Hd.shots_from_predictions(np.array([0] * 30 + [1] + [0] * 5 + [1] + [0] * 40, np.float32), 0.1)
    assert shots.tolist() == [[0, 36], [37, 76]]
