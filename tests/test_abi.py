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
