"""Device operators of the hot path (torch tensors in, torch tensors out).

Thin wrappers over the C ABI (include/svc.h): PyTorch is used only for device memory and
streams.  Every function requires a GPU and the built HIP library — nothing here has a
CPU path."""
import ctypes

import numpy as np
import torch

from . import _lib, weights as _weights
from ._lib import SvcParams

TAP_INPUT, TAP_FEAT4X, TAP_FEAT2X, TAP_FEAT1X, TAP_POSTCNN, TAP_DEC, TAP_PRE = range(7)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _need_cuda(t, dtype, name):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError('%s must be a contiguous CUDA tensor of dtype %s' % (name, dtype))


BLEND_NEXT, MAP_HELD = 1, 2          # bits of cluster_center_'s per-map flags (include/svc.h: SVC_BLEND_NEXT, SVC_MAP_HELD)


class Engine:
    """Owns one SvcHandle (weights + workspace) on one GPU.  Replaces the reference's module-level
    ``unisal_model`` singleton (smartVidCrop.py:77).  Not re-entrant, like the reference."""

    def __init__(self, state_dict=None, device=None, seed=0):
        if not torch.cuda.is_available():
            raise _lib.SvcError('no GPU visible: the SmartVidCrop hot path runs on the MI355X only')
        self.lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        if state_dict is None:
            state_dict = _weights.make_synthetic_state_dict(seed)
        blob = _weights.pack_blob(_weights.fold_state_dict(state_dict))
        import zlib
        self.weights_id = zlib.crc32(blob) & 0xffffffff        # identifies the checkpoint (smartVidCrop's feature cache keys on it)
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(blob, len(blob))
        _lib.check(self.lib.svc_create(buf, len(blob), self.device.index, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self.lib.svc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ingest down-scale -------------------------------------------------------------
    def resize_frames(self, frames, sh, sw):
        """uint8 [n,h,w,3] -> uint8 [n,sh,sw,3], cv2.resize(INTER_LINEAR) semantics."""
        _need_cuda(frames, torch.uint8, 'frames')
        n, h, w, c = frames.shape
        assert c == 3
        out = torch.empty((n, sh, sw, 3), dtype=torch.uint8, device=frames.device)
        _lib.check(self.lib.svc_resize_frames_u8(self._h, _ptr(frames), n, h, w, _ptr(out), sh, sw, _stream()))
        return out

    # -- saliency ------------------------------------------------------------------------
    def saliency(self, frames, out=None, threshold=0, census=None):
        """uint8 [n,h,w,3] RGB at saliency size -> uint8 [n,h,w] maps (frame-major); `out`: write into this tensor.
        threshold > 0: the maps come out thresholded (sc_threshold fused into the network's last kernel: the bytes of
        saliency() + threshold_(), one launch less).  census (with a threshold): int32 CUDA rows [n,4] to which the pixels
        of every frame's raw map at t - 1, t, t + 1 are ADDED (svc_saliency_census_u8; the caller zeroes the rows)."""
        _need_cuda(frames, torch.uint8, 'frames')
        n, h, w, c = frames.shape
        assert c == 3
        if out is None:
            out = torch.empty((n, h, w), dtype=torch.uint8, device=frames.device)
        else:
            _need_cuda(out, torch.uint8, 'out')
            assert tuple(out.shape) == (n, h, w)
        if census is not None:
            _need_cuda(census, torch.int32, 'census')
            assert threshold and tuple(census.shape) == (n, 4)
            _lib.check(self.lib.svc_saliency_census_u8(self._h, _ptr(frames), n, h, w, _ptr(out), int(threshold), _ptr(census), _stream()))
        elif threshold:
            _lib.check(self.lib.svc_saliency_thresholded_u8(self._h, _ptr(frames), n, h, w, _ptr(out), int(threshold), _stream()))
        else:
            _lib.check(self.lib.svc_saliency_u8(self._h, _ptr(frames), n, h, w, _ptr(out), _stream()))
        return out

    def tap(self, which, frame, shape):
        out = np.empty(int(np.prod(shape)), np.float32)
        _lib.check(self.lib.svc_debug_tap(self._h, which, frame, out.ctypes.data_as(ctypes.c_void_p), out.size))
        return out.reshape(shape)

    # -- tail ----------------------------------------------------------------------------
    def threshold_(self, maps, t):
        _need_cuda(maps, torch.uint8, 'maps')
        _lib.check(self.lib.svc_threshold_u8(self._h, _ptr(maps), maps.numel(), int(t), _stream()))
        return maps

    def cluster_center_(self, maps, blend_flags, CP, want_stats=False):
        """In place on thresholded uint8 [n,h,w] maps.  blend_flags: host sequence of n flags (or None): True / BLEND_NEXT =
        blend map i into map i+1 once it is final; MAP_HELD = not processed by this call (already final, or left for a
        later one), so a blend chain can be carried over between calls.  -> xy float64 [n,2] (NaN = None; unspecified for
        held maps) [, stats int32 [n,4]]."""
        _need_cuda(maps, torch.uint8, 'maps')
        n, h, w = maps.shape
        p = _lib.make_params(CP)
        xy = torch.empty((n, 2), dtype=torch.float64, device=maps.device)
        stats = torch.zeros((n, 4), dtype=torch.int32, device=maps.device) if want_stats else None
        flags = None
        if blend_flags is not None:
            flags = np.ascontiguousarray(np.asarray(blend_flags, np.uint8))
            assert flags.shape == (n,)
        _lib.check(self.lib.svc_cluster_center(
            self._h, _ptr(maps), n, h, w, flags.ctypes.data_as(ctypes.c_void_p) if flags is not None else None,
            ctypes.byref(p), _ptr(xy), _ptr(stats) if want_stats else None, _stream()))
        return (xy, stats) if want_stats else xy

    # -- measurement door (bench.py) -------------------------------------------------------
    KERNEL_CLASSES = ('resize', 'lanczos', 'stem', 'pw', 'dw', 'resample', 'smooth', 'threshold', 'compact',
                      'core', 'prim', 'finish')

    def profile_enable(self, kernel_class):
        """kernel_class: name from KERNEL_CLASSES, or None to switch event recording off."""
        k = -1 if kernel_class is None else self.KERNEL_CLASSES.index(kernel_class)
        _lib.check(self.lib.svc_profile_enable(self._h, k))

    def profile_read(self):
        """-> (total_ms, launches) of the profiled class since the last read; synchronises."""
        ms, cnt = ctypes.c_double(), ctypes.c_int()
        _lib.check(self.lib.svc_profile_read(self._h, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def profile_read_raw(self):
        """-> (raw_total_ms, empty_pair_ms, launches): the ingredients of profile_read's corrected sum."""
        raw, pair, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _lib.check(self.lib.svc_profile_read_raw(self._h, ctypes.byref(raw), ctypes.byref(pair), ctypes.byref(cnt)))
        return raw.value, pair.value, cnt.value

    def front_fused(self):
        """True when the last saliency call ran LANCZOS + features.0 + features.1 as the one kernel k_front."""
        return bool(self.lib.svc_front_fused(self._h))

    def threshold_census(self, reset=False):
        """-> dict(maps, below, at, above, pixels_per_grey_level): pixels of the un-thresholded maps at t - 1 / t / t + 1 over
        the maps this engine produced through saliency(threshold=t) since the last reset (svc_threshold_census: the regime
        diagnostic of the threshold; synchronises)."""
        out = (ctypes.c_uint64 * 4)()
        _lib.check(self.lib.svc_threshold_census(self._h, ctypes.cast(out, ctypes.c_void_p), 1 if reset else 0))
        maps, lo, at, hi = (int(v) for v in out)
        return dict(maps=maps, below=lo, at=at, above=hi, pixels_per_grey_level=((lo + at + hi) / (3.0 * maps) if maps else None))

    def matrix_pipe(self):
        """'f32' (fp32 MFMA) or 'bf16x6' (split-bf16 operands, six plane pairs on the bf16 MFMA): what the handle's 1x1
        convolutions run on (svc_matrix_pipe; environment SVC_MX when the engine is created)."""
        return 'bf16x6' if self.lib.svc_matrix_pipe(self._h) == 6 else 'f32'

    def argsort_u32(self, keys):
        """Test door: the device's emulation of numpy's default argsort on uint32 keys -> int32 order."""
        keys = np.ascontiguousarray(keys, np.uint32)
        out = np.empty(keys.size, np.int32)
        vp = ctypes.c_void_p
        _lib.check(self.lib.svc_debug_argsort_u32(self._h, keys.ctypes.data_as(vp), keys.size, out.ctypes.data_as(vp)))
        return out

    def cluster_state(self, frame, cap):
        pts = np.zeros(cap, np.uint32)
        core = np.zeros(cap, np.uint32)
        mst = np.zeros((cap, 3), np.uint32)
        labels = np.zeros(cap, np.int32)
        hdr = np.zeros(32, np.int32)
        vp = ctypes.c_void_p
        n = _lib.check(self.lib.svc_debug_cluster_state(self._h, frame, cap, pts.ctypes.data_as(vp),
                                                         core.ctypes.data_as(vp), mst.ctypes.data_as(vp),
                                                         labels.ctypes.data_as(vp), hdr.ctypes.data_as(vp)))
        m = min(n, cap)
        return dict(n=n, pts=pts[:m], core=core[:m], mst=mst[:max(m - 1, 0)], labels=labels[:m], hdr=hdr)


def iou_boxes(a, b):
    """IoU (inclusive +1 convention) of int32 box arrays [M,4] on the GPU -> float64 numpy [M].
    Accepts numpy arrays or CUDA tensors."""
    if not torch.cuda.is_available():
        raise _lib.SvcError('no GPU visible: svc_iou_i32 runs on the MI355X only')
    lib = _lib.load()
    ta = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, np.int32)).cuda()
    tb = b if torch.is_tensor(b) else torch.from_numpy(np.ascontiguousarray(b, np.int32)).cuda()
    _need_cuda(ta, torch.int32, 'a')
    _need_cuda(tb, torch.int32, 'b')
    assert ta.shape == tb.shape and ta.shape[-1] == 4
    m = ta.shape[0]
    out = torch.empty(m, dtype=torch.float64, device=ta.device)
    _lib.check(lib.svc_iou_i32(_ptr(ta), _ptr(tb), m, _ptr(out), _stream()))
    return out.cpu().numpy()
