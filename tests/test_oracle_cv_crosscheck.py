"""Second implementations for the OpenCV restatements in oracle/cv_ref.py (cv2 itself is not available offline and
the reference holds no resized / closed fixture, so these are cross-checks, not pins to cv2 binaries):

  CLOSE 5x5   scipy.ndimage grey dilation then erosion with the border ignored (constant 0 / 255 padding):
              bit for bit (smartVidCrop.py:1126-1128)
  INTER_LINEAR  torch's float bilinear (align_corners=False, no antialias) samples the same positions with the same
              edge clamping; OpenCV's 11-bit fixed-point weights may move a value by one grey level
              (smartVidCrop.py:333-335, :1080, :1158); the 2x case has the closed form (a+b+c+d+2)>>2
  INTER_NEAREST torch 'nearest' = floor(dst * scale), exact (smartVidCrop.py:1184)"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage

from oracle import cv_ref


def _maps(seed, h=140, w=250):
    rng = np.random.RandomState(seed)
    m = np.zeros((h, w), np.uint8)
    ys, xs = np.mgrid[0:h, 0:w]
    for _ in range(rng.randint(1, 5)):
        cy, cx, ry, rx = rng.randint(0, h), rng.randint(0, w), rng.randint(3, 20), rng.randint(3, 30)
        blob = (((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2) < 1
        m[blob] = rng.randint(120, 256, blob.sum())
    m[rng.rand(h, w) < 0.01] = rng.randint(120, 256)
    m[rng.rand(h, w) < 0.03] = 0
    return m


def test_close_5x5_equals_scipy_grey_morphology():
    for seed in range(12):
        m = _maps(seed) if seed < 10 else (np.random.RandomState(seed).randint(0, 256, (37, 61)).astype(np.uint8))
        d = ndimage.grey_dilation(m, size=(5, 5), mode='constant', cval=0)          # border samples never win a maximum
        ref = ndimage.grey_erosion(d, size=(5, 5), mode='constant', cval=255)       # ... nor a minimum
        assert np.array_equal(cv_ref.morph_close_5x5(m), ref), seed
    one = np.zeros((9, 9), np.uint8)
    one[0, 0] = 200                                                                 # a corner pixel survives its own closing
    assert cv_ref.morph_close_5x5(one)[0, 0] == 200 and cv_ref.morph_close_5x5(one).sum() == 200


def _torch_bilinear(img, oh, ow):
    t = torch.from_numpy(img.astype(np.float64))
    t = t.permute(2, 0, 1)[None] if img.ndim == 3 else t[None, None]
    o = F.interpolate(t, size=(oh, ow), mode='bilinear', align_corners=False, antialias=False)[0]
    return (o.permute(1, 2, 0) if img.ndim == 3 else o[0]).numpy()


def test_inter_linear_geometry_against_float_bilinear():
    rng = np.random.RandomState(1)
    for (h, w, oh, ow) in [(360, 640, 140, 250), (1080, 1920, 140, 250), (480, 640, 187, 250), (640, 360, 250, 140), (37, 53, 20, 31),
                           (35, 62, 140, 250)]:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        got = cv_ref.resize_linear_u8(img, oh, ow).astype(np.float64)
        ref = _torch_bilinear(img, oh, ow)
        assert np.abs(got - ref).max() <= 1.0, (h, w)                  # fixed point vs float: within one grey level
        assert (np.abs(got - np.rint(ref)) == 0).mean() > 0.8, (h, w)  # and the rounded float value on most pixels (87 % on white noise)
    m = _maps(3)
    small = cv_ref.resize_linear_factor_u8(m, 0.25)                    # best settings: 140x250 -> 35x62 (cvRound)
    assert small.shape == (35, 62)
    # fx = 1/4 is passed as the scale (not 250/62): torch's scale_factor path samples the same positions
    t = torch.from_numpy(m.astype(np.float64))[None, None]
    ref4 = F.interpolate(t, scale_factor=0.25, mode='bilinear', align_corners=False, recompute_scale_factor=False)[0, 0].numpy()
    assert ref4.shape[0] == 35 and np.abs(small[:, :ref4.shape[1]].astype(float) - ref4).max() <= 1.0


def test_inter_linear_half_size_closed_form():
    """Scale exactly 2: both taps weigh 1024/2048 and the fixed-point pipeline collapses to (a+b+c+d+2)>>2, which is
    also what OpenCV's own INTER_AREA fast path (taken for INTER_LINEAR at integer scale 2) computes."""
    rng = np.random.RandomState(2)
    img = rng.randint(0, 256, (64, 96)).astype(np.uint8)
    a = img.astype(np.int64)
    ref = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert np.array_equal(cv_ref.resize_linear_u8(img, 32, 48), ref.astype(np.uint8))


def test_inter_nearest_equals_torch_nearest():
    for seed in range(4):
        m = _maps(seed)
        got = cv_ref.resize_nearest_factor_u8(m, 0.25)
        t = torch.from_numpy(m.astype(np.float32))[None, None]
        ref = F.interpolate(t, scale_factor=0.25, mode='nearest', recompute_scale_factor=False)[0, 0].numpy().astype(np.uint8)
        assert got.shape == (35, 62)
        assert np.array_equal(got[:ref.shape[0], :ref.shape[1]], ref[:35, :62])


def _c_port():
    """tests/native/cv_resize_port.c (a separate restatement in the shape of OpenCV's resize.cpp), compiled here."""
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, 'oracle', '_build', 'libcv_resize_port.so')
    src = os.path.join(root, 'tests', 'native', 'cv_resize_port.c')
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-o', so, src, '-lm'])
    lib = ctypes.CDLL(so)
    lib.cv_resize_linear_u8.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]

    def resize(img, oh, ow, sx=0.0, sy=0.0):
        img = np.ascontiguousarray(img, np.uint8)
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty((oh, ow) if img.ndim == 2 else (oh, ow, cn), np.uint8)
        assert lib.cv_resize_linear_u8(img.ctypes.data, img.shape[0], img.shape[1], cn, out.ctypes.data, oh, ow, sx, sy) == 0
        return out
    return resize


def test_inter_linear_bit_equal_to_the_c_port_of_opencvs_resize():
    """oracle/cv_ref.resize_linear_u8 against tests/native/cv_resize_port.c on the sizes the path uses: the ingest
    down-scale from 640x360, 1080p and 4K, odd sizes (edge columns at and beyond xmax), and the best-settings pair
    140x250 -> 35x62 -> 140x250 with the explicit factor-4 scale of cv2.resize(fx=, fy=)."""
    c_resize = _c_port()
    rng = np.random.RandomState(3)
    for (h, w, oh, ow, cn) in [(360, 640, 140, 250, 3), (1080, 1920, 140, 250, 3), (2160, 3840, 140, 250, 3),
                               (480, 640, 187, 250, 3), (37, 53, 20, 31, 3), (20, 31, 37, 53, 1), (5, 7, 11, 13, 3),
                               (140, 250, 141, 251, 1), (90, 160, 140, 250, 3)]:
        img = rng.randint(0, 256, (h, w, cn) if cn > 1 else (h, w)).astype(np.uint8)
        assert np.array_equal(cv_ref.resize_linear_u8(img, oh, ow), c_resize(img, oh, ow)), (h, w, oh, ow)
    for seed in range(6):
        m = _maps(seed)
        small = cv_ref.resize_linear_factor_u8(m, 1.0 / 4)                      # smartVidCrop.py:1080 (fx = fy = 1 / factor)
        assert small.shape == (35, 62)
        assert np.array_equal(small, c_resize(m, 35, 62, 4.0, 4.0))
        back = cv_ref.resize_linear_u8(small, 140, 250)                          # :1158 ((initW, initH))
        assert np.array_equal(back, c_resize(small, 140, 250))
        # the upscale's last columns lie beyond xmax (plain copies of the last source column)
        assert np.array_equal(back[:, -2:], c_resize(small, 140, 250)[:, -2:])
