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
