"""The C-ABI library loads on a machine without a GPU and exports every symbol include/pbd.h declares.
No compute call is made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from partsbaseddetector_amd import build, _lib
    build.build_hip()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pbd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pbd_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from partsbaseddetector_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_version_and_kernel_names(lib):
    assert b"gfx950" in lib.pbd_version()
    from partsbaseddetector_amd import _lib
    for k, name in enumerate(_lib.KERNELS):
        assert lib.pbd_kernel_name(k).decode() == name


def test_create_fails_loudly_without_gpu(lib):
    """No CPU fallback: without a HIP device pbd_create must fail with PBD_ERR_HIP and say why."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from partsbaseddetector_amd import _lib, model as M
    flat = M.synthetic_tiny_model().flatten()
    cm = _lib.c_model(flat)
    cfg = _lib.CConfig(0, _lib.REAL_F32, _lib.CONV_EXACT, 1, 1024, None)
    h = C.c_void_p()
    rc = lib.pbd_create(C.byref(cm), C.byref(cfg), C.byref(h))
    assert rc == -3 and not h.value
    assert b"no CPU path" in lib.pbd_last_error(None)


def test_unsupported_real_type_is_reported(lib):
    from partsbaseddetector_amd import _lib, model as M
    flat = M.synthetic_tiny_model().flatten()
    cm = _lib.c_model(flat)
    cfg = _lib.CConfig(0, 7, _lib.CONV_EXACT, 1, 1024, None)
    h = C.c_void_p()
    assert lib.pbd_create(C.byref(cm), C.byref(cfg), C.byref(h)) == -2


def test_python_mirror_has_reference_method_names():
    from partsbaseddetector_amd import detector as d
    for cls, methods in {d.HOGFeatures: ["binsize", "nscales", "scales", "pyramid"],
                         d.SpatialConvolutionEngine: ["setFilters", "pdf"],
                         d.DynamicProgram: ["min", "argmin"],
                         d.PartsBasedDetector: ["distributeModel", "detect", "name"],
                         d.Candidate: ["score", "sort"]}.items():
        for m in methods:
            assert hasattr(cls, m), (cls, m)


def test_conv_tile_cover_is_complete_and_minimal():
    """Host logic of the convolution's mixed-shape tiling (pbd_capi.hip: cover_level), no GPU needed: every cell
    of a level is covered by a tile, and the number of lanes spent equals the optimum of the same dynamic
    program written independently here (strips of 32x8 / 16x16 / 8x32 tiles chosen over the columns)."""
    import ctypes as C
    import math
    import numpy as np
    from partsbaseddetector_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else _lib.load()
    fn = lib.pbd_debug_cover_level
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    shapes = [(32, 8), (16, 16), (8, 32)]
    rng = np.random.default_rng(4)
    cases = [(118, 158), (1, 1), (5, 7), (8, 32), (33, 17), (268, 478)] + [tuple(int(v) for v in rng.integers(1, 200, 2)) for _ in range(40)]
    for rows, cols in cases:
        buf = (C.c_int * (3 * 4096))()
        n = fn(rows, cols, buf, 4096)
        assert 0 < n <= 4096
        cover = np.zeros((rows, cols), np.int32)
        lanes = 0
        for i in range(n):
            k, y0, x0 = buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]
            tw, th = shapes[k]
            assert 0 <= y0 < rows and 0 <= x0 < cols and y0 % th == 0
            cover[y0:y0 + th, x0:x0 + tw] += 1
            lanes += tw * th
        assert cover.min() >= 1, (rows, cols)
        best = [0] + [10 ** 12] * cols
        for w in range(1, cols + 1):
            for tw, th in shapes:
                best[w] = min(best[w], best[max(w - tw, 0)] + tw * math.ceil(rows / th) * th)
        assert lanes == best[cols], (rows, cols, lanes, best[cols])


def test_no_exception_crosses_the_abi(lib):
    """include/pbd.h: "No exception crosses this ABI".  Every extern "C" body runs inside the same guard; its
    self-test entry point throws inside that guard (bad_alloc, a length_error from an absurd std::vector size,
    another std::exception) and must come back with a status code instead of std::terminate."""
    assert lib.pbd_debug_guard_selftest(0) == -6          # PBD_ERR_NOMEM
    assert b"memory" in lib.pbd_last_error(None)
    assert lib.pbd_debug_guard_selftest(1) == -6
    assert lib.pbd_debug_guard_selftest(2) == -1          # PBD_ERR_INVALID
    assert b"selftest" in lib.pbd_last_error(None)
    assert lib.pbd_debug_guard_selftest(3) == 0
