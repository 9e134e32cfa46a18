"""Oracle: the OpenCV operators on the hot path, restated in NumPy (K12, K15, ingest resize).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference call sites (smartVidCrop.py):
  :333-335, :633-635  cv2.resize(frame, (SAL_W, SAL_H), INTER_LINEAR)     ingest down-scale
  :1080, :1158        cv2.resize(sal_map, fx=fy=1/factor | (initW, initH), INTER_LINEAR)
  :1184               cv2.resize(sal_map, fx=fy=1/factor, INTER_NEAREST)
  :1127-1128          cv2.morphologyEx(sal_map, MORPH_CLOSE, ones(5,5))

OpenCV (opencv-python 4.2, README.md:85) is third-party, not vendored under
/root/reference and not installed here, so these follow OpenCV's published C
implementation for 8-bit images:

  * INTER_LINEAR on u8: src coordinate fx=(dx+0.5)*scale-0.5 evaluated in float32,
    floor + fraction, edge clamping, 11-bit fixed-point weights
    (saturate_cast<short>(w*2048), round-half-even), horizontal pass in int32, vertical
    pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
  * INTER_NEAREST: sx = min(floor(dx*scale), src-1).
  * dsize from fx: cvRound(src*fx) (round-half-even).
  * grey CLOSE = dilate then erode with a 5x5 rectangle anchored at the centre;
    samples outside the image are ignored (default morphology border value).

Parity status: UNPINNED against cv2 itself (no reference test holds resized or closed images and
cv2 is not available offline).  Cross-checked against second implementations in
tests/test_oracle_cv_crosscheck.py: CLOSE bit for bit against scipy.ndimage grey morphology,
INTER_NEAREST bit for bit against torch 'nearest', INTER_LINEAR within one grey level of torch's
float bilinear (same sample positions and clamping; the 11-bit weights are what differs),
exactly (a+b+c+d+2)>>2 at scale 2, and BIT FOR BIT against tests/native/cv_resize_port.c -- a separately written C
program in the shape of OpenCV's own code -- on the ingest sizes (640x360, 1080p, 4K -> 250x140), odd sizes and the
best-settings pair 140x250 -> 35x62 -> 140x250 including the columns at and beyond xmax.  Default settings use only
the down-scale (ingest, §8(f)-2) and CLOSE.

Which OpenCV code each expression restates (modules/imgproc/src/resize.cpp of OpenCV 4.x; no line numbers: the
sources are not available offline):
  _linear_coeffs      the INTER_LINEAR branch of cv::resize that fills xofs / ialpha and yofs / ibeta:
                      fx = (float)((dx + 0.5) * scale_x - 0.5); sx = cvFloor(fx); fx -= sx;
                      if (sx < ksize2 - 1) { xmin = dx + 1; if (sx < 0) fx = 0, sx = 0; }
                      if (sx + ksize2 >= ssize.width) { xmax = min(xmax, dx); if (sx >= ssize.width - 1) fx = 0, sx = ssize.width - 1; }
                      ialpha = saturate_cast<short>(cbuf[k] * INTER_RESIZE_COEF_SCALE)   (cvRound: round half to even)
  rows = ... (xmax)   HResizeLinear<uchar, int, short, INTER_RESIZE_COEF_SCALE>::operator(): D[dx] = S[sx] * a0 + S[sx + cn] * a1
                      for dx < xmax, D[dx] = S[sx] * ONE from xmax on
  s0 / s1 row fetch   resizeGeneric_Invoker: sy = clip(sy0 - ksize2 + 1 + k, 0, ssize.height - 1)
  out = ...           VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, INTER_RESIZE_COEF_BITS * 2>, VResizeLinearVec_32s8u>:
                      dst[x] = uchar(( ((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2)
  resize_*_factor     cv::resize with dsize empty: dsize = Size(saturate_cast<int>(ssize.width * inv_scale_x), ...) (cvRound),
                      scale_x = 1 / inv_scale_x = the factor itself
  resize_nearest      resizeNN: sx = min(cvFloor(x * ifx), ssize.width - 1)
  morph_close_5x5     cv::morphologyEx(MORPH_CLOSE) = dilate, erode with morphologyDefaultBorderValue() (borders never win)
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def cv_round(x):
    return int(np.rint(x))          # round half to even, like cvRound on SSE2 builds


def _fix(w):
    return int(np.clip(np.rint(np.float32(w * np.float32(COEF_SCALE))), -32768, 32767))


def _linear_coeffs(src, dst, scale, horizontal):
    """Offsets and 11-bit weights.  The horizontal table clamps at the image edges
    (fx=0 at the left, plain copy from xmax on); the vertical table keeps the raw
    floor/fraction and the row fetch clips the row index instead."""
    ofs = np.zeros(dst, np.int64)
    a = np.zeros((dst, 2), np.int64)
    xmax = dst
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if horizontal:
            if s < 0:
                f, s = np.float32(0), 0
            if s + 1 >= src:
                xmax = min(xmax, d)
                if s >= src - 1:
                    f, s = np.float32(0), src - 1
        ofs[d] = s
        a[d, 0] = _fix(np.float32(1.0) - f)
        a[d, 1] = _fix(f)
    return ofs, a, xmax


def resize_linear_u8(img, out_h, out_w, scale_x=None, scale_y=None):
    """cv2.resize(img, (out_w, out_h), interpolation=INTER_LINEAR) for u8 HW or HWC."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w, _ = img.shape
    sx = float(w) / out_w if scale_x is None else scale_x
    sy = float(h) / out_h if scale_y is None else scale_y
    xofs, alpha, xmax = _linear_coeffs(w, out_w, sx, True)
    yofs, beta, _ = _linear_coeffs(h, out_h, sy, False)
    src = img.astype(np.int64)
    x1 = np.minimum(xofs + 1, w - 1)
    # horizontal pass for every source row (fits int32)
    rows = src[:, xofs, :] * alpha[None, :, 0, None] + src[:, x1, :] * alpha[None, :, 1, None]
    if xmax < out_w:
        rows[:, xmax:, :] = src[:, xofs[xmax:], :] * COEF_SCALE
    s0 = rows[np.clip(yofs, 0, h - 1)]
    s1 = rows[np.clip(yofs + 1, 0, h - 1)]
    b0 = beta[:, 0][:, None, None]
    b1 = beta[:, 1][:, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def resize_linear_factor_u8(img, inv_factor):
    """cv2.resize(img, None, fx=inv_factor, fy=inv_factor, INTER_LINEAR)."""
    h, w = img.shape[:2]
    oh, ow = cv_round(h * inv_factor), cv_round(w * inv_factor)
    return resize_linear_u8(img, oh, ow, 1.0 / inv_factor, 1.0 / inv_factor)


def resize_nearest_factor_u8(img, inv_factor):
    """cv2.resize(img, None, fx=inv_factor, fy=inv_factor, INTER_NEAREST)."""
    h, w = img.shape[:2]
    oh, ow = cv_round(h * inv_factor), cv_round(w * inv_factor)
    ifx = 1.0 / inv_factor
    xs = np.minimum(np.floor(np.arange(ow) * ifx).astype(np.int64), w - 1)
    ys = np.minimum(np.floor(np.arange(oh) * ifx).astype(np.int64), h - 1)
    return img[ys][:, xs]


def _window_reduce(img, k, fn, fill):
    r = k // 2
    h, w = img.shape
    pad = np.full((h + 2 * r, w + 2 * r), fill, img.dtype)
    pad[r:r + h, r:r + w] = img
    out = pad[0:h, 0:w].copy()
    for dy in range(k):
        for dx in range(k):
            out = fn(out, pad[dy:dy + h, dx:dx + w])
    return out


def morph_close_5x5(img):
    """cv2.morphologyEx(img, MORPH_CLOSE, ones((5,5))) for a u8 HW image."""
    d = _window_reduce(img, 5, np.maximum, 0)
    return _window_reduce(d, 5, np.minimum, 255)
