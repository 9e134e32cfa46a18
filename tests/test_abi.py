"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/svc.h
declares; without a GPU svc_create reports an error instead of crashing."""
import ctypes
import os
import re

import torch

from retargetvid_amd import _lib, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'svc.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(svc_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 10 and set(names) == set(_lib.EXPORTS)
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), n


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.SvcParams) == 32
    assert [f[0] for f in _lib.SvcParams._fields_] == ['struct_size', 'hdbscan_min', 'hdbscan_min_samples', 'select_sum', 'op_close',
                                                       'clust_filt', 'resize_factor', 'com_km']


def test_abi_version_matches_header():
    text = open(os.path.join(ROOT, 'include', 'svc.h')).read()
    assert int(re.search(r'#define SVC_ABI_VERSION (\d+)', text).group(1)) == _lib.ABI_VERSION == _lib.load().svc_abi_version()


def test_create_reports_errors():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.svc_create(b'\0' * 64, 64, 0, ctypes.byref(h)) < 0 and b'magic' in lib.svc_last_error()
    if not torch.cuda.is_available():
        blob = weights.pack_blob(weights.fold_state_dict(weights.make_synthetic_state_dict(0)))
        buf = ctypes.create_string_buffer(blob, len(blob))
        assert lib.svc_create(buf, len(blob), 0, ctypes.byref(h)) < 0
        assert len(lib.svc_last_error()) > 0
    assert lib.svc_destroy(None) == 0


def test_host_stage_entries_check_their_arguments():
    """The svc_host_* entries (host memory, no GPU): a SvcTemporalParams compiled against another layout is rejected by its
    struct_size, null / inconsistent arguments give SVC_E_INVALID + a message instead of a crash, n = 0 is a no-op."""
    import numpy as np
    lib = _lib.load()
    assert ctypes.sizeof(_lib.SvcTemporalParams) == 40
    assert [f[0] for f in _lib.SvcTemporalParams._fields_] == ['struct_size', 'lp_filt', 'lp_taps', 'loess_filt', 'loess_degree', 'reserved',
                                                               'loess_w_secs', 'fr']
    vp = ctypes.c_void_p
    xy = np.array([[10.0, 20.0], [11.0, 21.0], [12.0, 22.0]])
    cx, cy = np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1])
    ti, seg, sel = np.array([0, 6, 11], np.int32), np.array([[0, 11]], np.int32), np.array([[0, 2]], np.int32)
    out = np.empty((4, 12))
    p = _lib.SvcTemporalParams(ctypes.sizeof(_lib.SvcTemporalParams), 0, 0, 1, 2, 0, 2.0, 30.0)
    args = lambda pp: (ctypes.byref(pp), None, None, None, cx.ctypes.data_as(vp), cy.ctypes.data_as(vp), 3, ti.ctypes.data_as(vp),
                       seg.ctypes.data_as(vp), sel.ctypes.data_as(vp), 1, 12, out[0].ctypes.data_as(vp), out[1].ctypes.data_as(vp),
                       out[2].ctypes.data_as(vp), out[3].ctypes.data_as(vp))
    assert lib.svc_host_temporal(*args(p)) == 12                          # 12 frames produced (linear interpolation of 3 samples)
    assert out[0][0] == 10.0 and out[0][6] == 11.0 and out[0][11] == 12.0
    stale = _lib.SvcTemporalParams(32, 0, 0, 1, 2, 0, 2.0, 30.0)          # a binding built against a shorter struct
    assert lib.svc_host_temporal(*args(stale)) == -1 and b'struct_size' in lib.svc_last_error()
    p_lp = _lib.SvcTemporalParams(ctypes.sizeof(_lib.SvcTemporalParams), 1, 6, 1, 2, 0, 2.0, 30.0)     # low-pass on, no coefficients
    assert lib.svc_host_temporal(*args(p_lp)) == -1
    assert lib.svc_host_loess(None, 0, 5, 2, None) == 0                   # n = 0: a no-op
    assert lib.svc_host_loess(None, 4, 5, 2, None) == -1
    assert lib.svc_host_boxes(None, None, 0, 640, 360, 250, 140, 120, 360, None, None, None, None) == 0
    assert lib.svc_host_interp_segment(None, None, None, 3, 3, None, None) == -1 and b'svc_host_interp_segment' in lib.svc_last_error()
    y = np.arange(8.0)
    o = np.empty(8)
    assert lib.svc_host_savgol(y.ctypes.data_as(vp), 8, 4, 2, o.ctypes.data_as(vp)) == -1          # even window: savgol_filter raises
    assert lib.svc_host_focus_stability(None, None, 0, None, 140, 250, 30.0, 6, 1.0, 60.0, 1.5, None, None) == 0
    assert lib.svc_host_focus_stability(None, None, 3, None, 140, 250, 30.0, 6, 1.0, 60.0, 1.5, None, None) == -1
