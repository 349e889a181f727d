"""ctypes binding of the C ABI in include/pbd.h (libpbd_hip.so).

Fails loudly when the library is missing or cannot be loaded: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBD_LIB", os.path.join(HERE, "libpbd_hip.so"))
MAX_LEVELS = 128

PBD_OK = 0
STATUS = {0: "PBD_OK", -1: "PBD_ERR_INVALID", -2: "PBD_ERR_UNSUPPORTED", -3: "PBD_ERR_HIP", -4: "PBD_ERR_CAPACITY",
          -5: "PBD_ERR_STATE", -6: "PBD_ERR_NOMEM"}
REAL_F32, REAL_F64 = 0, 1
CONV_EXACT, CONV_FMA, CONV_MFMA, CONV_MFMA_F16 = 0, 1, 2, 3
STAGE_FEATURES, STAGE_RESPONSES, STAGE_ROOTV, STAGE_ROOTI = 0, 1, 2, 3
# cv::Mat::depth() codes of the image depths HOGFeatures::pyramid accepts (src/HOGFeatures.cpp:136-146)
DEPTH_CODE = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 2, np.dtype(np.float32): 5, np.dtype(np.float64): 6}
KERNELS = ["k_resize", "k_pyrdown", "k_hog_hist", "k_hog_feat", "k_conv", "k_dt_rows", "k_dt_cols", "k_dp_combine",
           "k_dp_root", "k_argmin"]

# every symbol include/pbd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "pbd_create", "pbd_destroy", "pbd_last_error", "pbd_version", "pbd_candidate_stride", "pbd_binsize",
    "pbd_pyramid_plan", "pbd_set_level_shard", "pbd_features_pyramid", "pbd_get_pyramid_image", "pbd_conv_set_filters", "pbd_conv_pdf",
    "pbd_num_ptr_slots", "pbd_ptr_slot", "pbd_dp_min", "pbd_dp_argmin", "pbd_detect", "pbd_detect_batch",
    "pbd_detect_batch_device", "pbd_detect_typed", "pbd_detect_batch_submit", "pbd_detect_batch_wait",
    "pbd_detect_batch_device_submit", "pbd_detect_batch_device_out", "pbd_argmin_device_out", "pbd_stream", "pbd_get_stage", "pbd_profile_enable", "pbd_profile_reset", "pbd_profile_read",
    "pbd_kernel_name", "pbd_synchronize",
]


class PbdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class CModel(C.Structure):
    _fields_ = [
        ("ncomponents", C.c_int), ("nfilters", C.c_int), ("flen", C.c_int),
        ("filter_ksize", C.POINTER(C.c_int)), ("filter_offset", C.POINTER(C.c_int64)),
        ("filters_f32", C.POINTER(C.c_float)), ("filters_f64", C.POINTER(C.c_double)),
        ("nbias", C.c_int), ("biasw", C.POINTER(C.c_float)),
        ("ndefs", C.c_int), ("defw", C.POINTER(C.c_float)), ("anchors", C.POINTER(C.c_int)),
        ("part_offset", C.POINTER(C.c_int)), ("parentid", C.POINTER(C.c_int)),
        ("mix_offset", C.POINTER(C.c_int)), ("filterid", C.POINTER(C.c_int)),
        ("biasid", C.POINTER(C.c_int)), ("defid", C.POINTER(C.c_int)),
        ("thresh", C.c_float), ("sbin", C.c_int), ("interval", C.c_int), ("norient", C.c_int),
    ]


class CConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("real_type", C.c_int), ("conv_mode", C.c_int), ("max_batch", C.c_int),
                ("max_candidates", C.c_int), ("stream", C.c_void_p)]


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


_LIB = None


def load():
    """Load libpbd_hip.so (building it first is __graft_entry__.build()'s job)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -m partsbaseddetector_amd.build` "
                          "(the HIP extension is required; there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.pbd_last_error.restype = C.c_char_p
    lib.pbd_last_error.argtypes = [C.c_void_p]
    lib.pbd_version.restype = C.c_char_p
    lib.pbd_kernel_name.restype = C.c_char_p
    lib.pbd_create.argtypes = [C.POINTER(CModel), C.POINTER(CConfig), C.POINTER(C.c_void_p)]
    lib.pbd_destroy.argtypes = [C.c_void_p]
    lib.pbd_destroy.restype = None
    for name in ("pbd_candidate_stride", "pbd_binsize", "pbd_num_ptr_slots", "pbd_synchronize", "pbd_profile_reset"):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.pbd_ptr_slot.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pbd_set_level_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pbd_pyramid_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_float)]
    lib.pbd_features_pyramid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                         C.POINTER(C.c_void_p)]
    lib.pbd_get_pyramid_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pbd_conv_set_filters.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.pbd_conv_pdf.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_void_p)]
    lib.pbd_dp_min.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)] + [C.POINTER(C.c_void_p)] * 6
    lib.pbd_dp_argmin.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.pbd_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                               C.POINTER(C.c_int)]
    lib.pbd_detect_typed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_int,
                                     C.POINTER(C.c_int)]
    lib.pbd_detect_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_size_t,
                                     C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.pbd_detect_batch_submit.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_size_t]
    lib.pbd_detect_batch_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.pbd_detect_batch_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_int, C.POINTER(C.c_int)]
    lib.pbd_detect_batch_device_submit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.pbd_detect_batch_device_out.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.pbd_argmin_device_out.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.pbd_stream.argtypes = [C.c_void_p]
    lib.pbd_stream.restype = C.c_void_p
    lib.pbd_get_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.pbd_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.pbd_profile_read.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.pbd_kernel_name.argtypes = [C.c_int]
    _LIB = lib
    return lib


def c_model(flat) -> CModel:
    m = CModel()
    m.ncomponents, m.nfilters, m.flen = flat.ncomponents, flat.nfilters, flat.flen
    m.filter_ksize = ptr(flat.filter_ksize, C.c_int)
    m.filter_offset = ptr(flat.filter_offset, C.c_int64)
    m.filters_f32 = ptr(flat.filters_f32, C.c_float)
    m.filters_f64 = ptr(flat.filters_f64, C.c_double)
    m.nbias, m.biasw = len(flat.biasw), ptr(flat.biasw, C.c_float)
    m.ndefs, m.defw, m.anchors = len(flat.defw), ptr(flat.defw, C.c_float), ptr(flat.anchors, C.c_int)
    m.part_offset, m.parentid = ptr(flat.part_offset, C.c_int), ptr(flat.parentid, C.c_int)
    m.mix_offset, m.filterid = ptr(flat.mix_offset, C.c_int), ptr(flat.filterid, C.c_int)
    m.biasid, m.defid = ptr(flat.biasid, C.c_int), ptr(flat.defid, C.c_int)
    m.thresh, m.sbin, m.interval, m.norient = flat.thresh, flat.sbin, flat.interval, flat.norient
    m._keep = flat
    return m


def ptr_array(arrays):
    """(void*[]) over a list of numpy arrays; returns (ctypes array, keepalive)."""
    arr = (C.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ctypes.data if a is not None and a.size else None
    return arr
