"""Oracle: Pillow's 8-bit LANCZOS resampler restated in NumPy (K0).

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference resizes each saliency-size frame to the network input size with
``transforms.Resize(out_size, interpolation=PIL.Image.LANCZOS)``
(3rd_party_libs/unisal/unisal/data.py:1281-1294), i.e. ``PIL.Image.resize``.
The arithmetic lives in Pillow (third-party, ``src/libImaging/Resample.c``, not
vendored under /root/reference; the reference pins no Pillow version,
README.md:83-92).  Published algorithm restated here:

  * a=3 windowed sinc; support = 3 * max(1, in/out); ksize = ceil(support)*2+1
  * per output index: centre=(i+0.5)*scale, xmin=int(centre-support+0.5) clipped
    to 0, xmax=int(centre+support+0.5) clipped to in_size; weights normalised in
    float64, then converted to fixed point with 22 fractional bits, rounding
    half away from zero
  * horizontal pass first, then vertical; each pass accumulates in int32
    starting from 1<<21, shifts right by 22 and clips to [0,255] (u8
    intermediate between the passes)

Pinned: tests/golden/lanczos_*.npz hold outputs of the container's Pillow
(12.2.0) on seeded inputs (tools/make_golden_lanczos.py);
tests/test_oracle_lanczos.py requires bit-exact equality.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3.0)
    return 0.0


def precompute_coeffs(in_size, out_size):
    """-> (bounds[out,2] int32 (xmin, count), coeffs[out,ksize] int32 fixed-point)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            w = _lanczos((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS),
                     0.5 + kk * (1 << PRECISION_BITS))
    return bounds, np.trunc(fixed).astype(np.int32), ksize


def _resample_axis0(img, out_size):
    """Resample along axis 0 of an array [in, ...] u8 -> [out, ...] u8."""
    in_size = img.shape[0]
    bounds, coeffs, ksize = precompute_coeffs(in_size, out_size)
    src = img.astype(np.int64)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        k = coeffs[xx, :cnt].astype(np.int64)
        acc = np.tensordot(k, src[xmin:xmin + cnt], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_lanczos_u8(img_hwc, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), LANCZOS) for an HWC u8 image."""
    img = np.ascontiguousarray(img_hwc)
    if out_w != img.shape[1]:
        img = np.ascontiguousarray(
            _resample_axis0(np.ascontiguousarray(img.transpose(1, 0, 2)), out_w).transpose(1, 0, 2))
    if out_h != img.shape[0]:
        img = _resample_axis0(img, out_h)
    return img
