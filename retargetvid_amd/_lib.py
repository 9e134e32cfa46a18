"""ctypes binding of libsvc_hip.so (C ABI: include/svc.h).

There is deliberately no CPU fallback: if the shared library has not been built
(`make -C retargetvid_amd/csrc`, or `python -c "import __graft_entry__ as g; g.build()"`)
importing any device op raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SVC_LIB') or os.path.join(_HERE, 'libsvc_hip.so')      # SVC_LIB: another build of the same ABI (A/B runs)

ABI_VERSION = 5          # include/svc.h SVC_ABI_VERSION this binding was written against

EXPORTS = ('svc_last_error', 'svc_abi_version', 'svc_create', 'svc_destroy', 'svc_resize_frames_u8', 'svc_saliency_u8',
           'svc_threshold_u8', 'svc_cluster_center', 'svc_iou_i32', 'svc_debug_cluster_state', 'svc_debug_tap', 'svc_front_fused', 'svc_matrix_pipe', 'svc_threshold_census', 'svc_debug_round_plan', 'svc_transnet_load', 'svc_transnet_predict', 'svc_transnet_matrix_pipe',
           'svc_debug_argsort_u32',
           'svc_profile_enable', 'svc_profile_read', 'svc_profile_read_raw',
           'svc_host_fill_empty_centres', 'svc_host_interp_segment', 'svc_host_lowpass', 'svc_host_loess', 'svc_host_savgol',
           'svc_host_temporal', 'svc_host_boxes', 'svc_host_focus_stability', 'svc_saliency_thresholded_u8',
           'svc_saliency_census_u8', 'svc_transnet_predict_rows', 'svc_transnet_config_get', 'svc_transnet_config_set')


class SvcParams(ctypes.Structure):
    _fields_ = [('struct_size', ctypes.c_uint32), ('hdbscan_min', ctypes.c_int32), ('hdbscan_min_samples', ctypes.c_int32),
                ('select_sum', ctypes.c_int32), ('op_close', ctypes.c_int32), ('clust_filt', ctypes.c_int32),
                ('resize_factor', ctypes.c_int32), ('com_km', ctypes.c_int32)]


class SvcTemporalParams(ctypes.Structure):
    _fields_ = [('struct_size', ctypes.c_uint32), ('lp_filt', ctypes.c_int32), ('lp_taps', ctypes.c_int32),
                ('loess_filt', ctypes.c_int32), ('loess_degree', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('loess_w_secs', ctypes.c_double), ('fr', ctypes.c_double)]


def make_params(CP):
    """SvcParams from a crop-parameter dict (sc_init_crop_params keys, smartVidCrop.py:132-209)."""
    factor = CP.get('resize_factor', 1.0)
    if float(factor) != int(factor) or (int(factor) != 1 and CP.get('resize_type', 1) != 1):
        raise NotImplementedError('resize_factor must be an integer and resize_type 1 (bilinear)')
    return SvcParams(ctypes.sizeof(SvcParams), int(CP['hdbscan_min']), int(CP['hdbscan_min_samples'] or 0),
                     int(CP['select_sum']), int(bool(CP['op_close'])), int(bool(CP['clust_filt'])), int(factor),
                     int(bool(CP.get('com_km', True))))


class SvcError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library (once).  Raises SvcError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise SvcError('HIP library %s not found: build it with `make -C %s/csrc` '
                       '(there is no CPU fallback)' % (LIB_PATH, _HERE))
    # PyTorch first: its wheel bundles the HIP runtime (libamdhip64) the process must share -- device memory and streams come from
    # torch.  Loaded the other way round the library would bind the system's /opt/rocm runtime, and on the GPU boxes of this pool that
    # second runtime reports "no ROCm-capable device" (seen in round 6 with `python __graft_entry__.py smoke`: build() loaded the
    # library before anything had imported torch).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.svc_last_error.restype = ctypes.c_char_p
    lib.svc_last_error.argtypes = []
    lib.svc_abi_version.argtypes = []
    lib.svc_abi_version.restype = ctypes.c_int
    if lib.svc_abi_version() != ABI_VERSION:
        raise SvcError('%s has ABI version %d, this binding expects %d: rebuild the library'
                       % (LIB_PATH, lib.svc_abi_version(), ABI_VERSION))
    lib.svc_create.argtypes = [vp, sz, i32, ctypes.POINTER(vp)]
    lib.svc_destroy.argtypes = [vp]
    lib.svc_resize_frames_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, i32, vp]
    lib.svc_saliency_u8.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.svc_saliency_thresholded_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp]
    lib.svc_threshold_u8.argtypes = [vp, vp, sz, i32, vp]
    lib.svc_cluster_center.argtypes = [vp, vp, i32, i32, i32, vp, ctypes.POINTER(SvcParams), vp, vp, vp]
    lib.svc_iou_i32.argtypes = [vp, vp, sz, vp, vp]
    lib.svc_debug_cluster_state.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.svc_debug_tap.argtypes = [vp, i32, i32, vp, sz]
    lib.svc_front_fused.argtypes = [vp]
    lib.svc_matrix_pipe.argtypes = [vp]
    lib.svc_transnet_matrix_pipe.argtypes = [vp]
    lib.svc_threshold_census.argtypes = [vp, vp, i32]
    lib.svc_debug_round_plan.argtypes = [vp, i32, vp, vp]
    lib.svc_transnet_load.argtypes = [vp, vp, sz]
    lib.svc_transnet_predict.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.svc_transnet_predict_rows.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.svc_transnet_config_get.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    lib.svc_transnet_config_set.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    lib.svc_saliency_census_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.svc_debug_argsort_u32.argtypes = [vp, vp, i32, vp]
    lib.svc_profile_enable.argtypes = [vp, i32]
    lib.svc_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    lib.svc_profile_read_raw.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    lib.svc_host_fill_empty_centres.argtypes = [vp, vp, i32, vp, i32]
    lib.svc_host_interp_segment.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.svc_host_lowpass.argtypes = [vp, vp, vp, i32, vp, i32, vp]
    lib.svc_host_loess.argtypes = [vp, i32, i32, i32, vp]
    lib.svc_host_savgol.argtypes = [vp, i32, i32, i32, vp]
    lib.svc_host_temporal.argtypes = [ctypes.POINTER(SvcTemporalParams), vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.svc_host_boxes.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    f64 = ctypes.c_double
    lib.svc_host_focus_stability.argtypes = [vp, vp, i32, vp, i32, i32, f64, i32, f64, f64, f64, vp, vp]
    for name in EXPORTS:
        if name not in ('svc_last_error', 'svc_abi_version'):
            getattr(lib, name).restype = i32
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        raise SvcError('svc call failed (%d): %s' % (rc, load().svc_last_error().decode(errors='replace')))
    return rc
