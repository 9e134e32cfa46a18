"""-m gpu: the reference-side ctypes binding that INTEGRATION.md documents (option B) is executed as written:
the ``svc_binding.py`` block is extracted from the markdown, run against the built library, and its results
are compared with the packaged operators (retargetvid_amd.unisal_handler / ops) on the same inputs.  A stale
binding (older SvcParams layout) must be rejected by the library, not read past its end."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from retargetvid_amd import _lib, smartVidCrop as S, synth, unisal_handler, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'```python\n(# --- svc_binding\.py.*?)```', text, flags=re.S)
    assert m, 'INTEGRATION.md no longer holds the svc_binding.py block'
    return m.group(1)


def test_stub_declares_the_header_layout():
    """CPU part: the documented struct has the fields of include/svc.h in order (no GPU needed to see a drift)."""
    src = _stub_source()
    hdr = open(os.path.join(ROOT, 'include', 'svc.h')).read()
    body = re.search(r'typedef struct SvcParams \{(.*?)\} SvcParams;', hdr, flags=re.S).group(1)
    fields = re.findall(r'\b(?:u?int32_t)\s+(\w+);', body)
    assert fields == [f[0] for f in _lib.SvcParams._fields_]
    for f in fields:
        assert "'%s'" % f in src, f
    assert 'svc_abi_version() == %d' % _lib.ABI_VERSION in src


@pytest.mark.gpu
def test_documented_binding_runs_and_matches_the_package(engine, synthetic_sd):
    os.environ['SVC_LIB'] = _lib.LIB_PATH
    mod = types.ModuleType('svc_binding')
    exec(compile(_stub_source(), 'INTEGRATION.md:svc_binding.py', 'exec'), mod.__dict__)
    blob = weights.pack_blob(weights.fold_state_dict(synthetic_sd))
    h = mod.create(blob, 0)
    frames = synth.blob_frames(9, 140, 250, seed=5)
    smaps = mod.saliency(h, frames)                                               # [H,W,n] like unisal_handler.py:85-86
    ref = unisal_handler.predictions_from_memory_nuint8_np(engine, frames, [], '')
    assert smaps.shape == (140, 250, 9) and np.array_equal(smaps, ref)
    for CP in (S.sc_init_crop_params(), S.sc_init_crop_params(use_best_settings=True)):
        cuts = {0, 4, 8}
        filt, dx, dy = mod.cluster_and_centres(h, smaps.copy(), cuts, CP)
        d = torch.from_numpy(np.ascontiguousarray(smaps.transpose(2, 0, 1))).cuda()
        engine.threshold_(d, CP['t_threshold'])
        flags = np.array([i < 9 - 2 and any(x in cuts for x in (i - 1, i, i + 1)) for i in range(9)], np.uint8)
        xy = engine.cluster_center_(d, flags, CP).cpu().numpy()
        assert np.array_equal(filt, d.permute(1, 2, 0).cpu().numpy())
        for i in range(9):
            assert (dx[i] is None and np.isnan(xy[i, 0])) or (dx[i] == xy[i, 0] and dy[i] == xy[i, 1])
    _lib.load().svc_destroy(h)


@pytest.mark.gpu
def test_stale_binding_is_rejected_not_over_read(engine):
    """The round-1 layouts (five or six int32, no size member) must come back as SVC_E_INVALID."""
    lib = _lib.load()

    class Old5(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in ('hdbscan_min', 'hdbscan_min_samples', 'select_sum', 'op_close', 'clust_filt')]

    class Old6(ctypes.Structure):
        _fields_ = Old5._fields_ + [('resize_factor', ctypes.c_int32)]

    m = torch.zeros((1, 140, 250), dtype=torch.uint8, device='cuda')
    xy = torch.empty((1, 2), dtype=torch.float64, device='cuda')
    vp = ctypes.c_void_p
    class V2(ctypes.Structure):                      # ABI 2: struct_size = 28 and no com_km
        _fields_ = [('struct_size', ctypes.c_uint32)] + Old6._fields_

    for old in (Old5(26, 0, 2, 1, 1), Old6(26, 0, 2, 1, 1, 1), V2(28, 26, 0, 2, 1, 1, 1)):
        fn = lib.svc_cluster_center
        saved = fn.argtypes
        fn.argtypes = saved[:6] + [vp] + saved[7:]
        try:
            rc = fn(engine._h, vp(m.data_ptr()), 1, 140, 250, None, ctypes.cast(ctypes.byref(old), vp), vp(xy.data_ptr()), None, None)
        finally:
            fn.argtypes = saved
        assert rc == -1 and b'struct_size' in lib.svc_last_error()


@pytest.mark.gpu
def test_documented_transnet_binding_runs_and_matches_the_package(engine):
    """the `svc_transnet.py` block, executed as written against the built library."""
    from retargetvid_amd import transnetv1_handler as Hd
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'```python\n(# svc_transnet\.py.*?)```', text, flags=re.S)
    assert m, 'INTEGRATION.md no longer holds the svc_transnet.py block'
    mod = types.ModuleType('svc_transnet')
    exec(compile(m.group(1), 'INTEGRATION.md:svc_transnet.py', 'exec'), mod.__dict__)
    sd = weights.make_transnet_state_dict(2)
    lib = _lib.load()
    net = mod.ShotTransNetHIP(lib, engine._h, np.ascontiguousarray(weights.pack_transnet_blob(sd)))
    fr = np.random.RandomState(3).randint(0, 256, (2, 30, 27, 48, 3)).astype(np.uint8)
    got = net.predict_raw(fr)
    ref_net = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=sd, engine=engine)
    assert got.shape == (2, 30) and np.array_equal(got, ref_net.predict_raw(fr))
