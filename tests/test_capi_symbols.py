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


def test_conv_strip_sequence_tiles_cover_every_position_once():
    """Host logic of the exact convolution's tiles (pbd_capi.hip: build_seg_tiles), no GPU needed: the strips of four rows
    of every level of every frame form one sequence of positions; every position is in exactly one tile, a tile has at
    most 64 positions in at most three runs (each inside one strip), runs of a tile are consecutive in the sequence, and the
    tile count equals an independent greedy count.  640x480: 569 tiles per frame (549.7 = no idle lane at all)."""
    import ctypes as C
    import numpy as np
    from partsbaseddetector_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    fn = lib.pbd_debug_seg_tiles
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int]

    def tiles_of(levels, nb):
        rows = np.array([h for h, w in levels], np.int32)
        cols = np.array([w for h, w in levels], np.int32)
        cap = 200000
        buf = np.zeros(16 * cap, np.int32)
        n = fn(len(levels), rows.ctypes.data_as(C.POINTER(C.c_int)), cols.ctypes.data_as(C.POINTER(C.c_int)), nb,
               buf.ctypes.data_as(C.POINTER(C.c_int)), cap)
        assert 0 <= n <= cap
        return buf[:16 * n].reshape(n, 16)

    def greedy(levels, nb):
        tiles = lanes = segs = 0
        for f in range(nb):
            for h, w in levels:
                if h * w == 0:
                    continue
                for st in range((h + 3) // 4):
                    x = 0
                    while x < w:
                        if segs == 3 or lanes == 64:
                            tiles, lanes, segs = tiles + 1, 0, 0
                        take = min(w - x, 64 - lanes)
                        lanes, segs, x = lanes + take, segs + 1, x + take
        return tiles + (1 if lanes else 0)

    rng = np.random.default_rng(3)
    cases = [([(118, 158), (110, 147), (5, 7), (0, 9), (3, 2), (1, 1)], 3), ([(4, 64)], 2), ([(1, 200)], 1), ([(9, 1)], 5)]
    cases += [([tuple(int(v) for v in rng.integers(0, 90, 2)) for _ in range(int(rng.integers(1, 8)))], int(rng.integers(1, 4))) for _ in range(20)]
    for levels, nb in cases:
        t = tiles_of(levels, nb)
        seq = []                                        # the sequence of (frame, level, strip, x) in tile order
        for r in t:
            ns = int(r[0])
            assert 1 <= ns <= 3 and 0 < int(r[1:1 + ns].sum()) <= 64 and not r[1 + ns:4].any()
            for k in range(ns):
                f, l, st, x0 = (int(v) for v in r[4 + 4 * k: 8 + 4 * k])
                h, w = levels[l]
                assert 0 <= f < nb and 0 <= st < (h + 3) // 4 and 0 <= x0 and x0 + r[1 + k] <= w
                seq += [(f, l, st, x) for x in range(x0, x0 + int(r[1 + k]))]
        want = [(f, l, st, x) for f in range(nb) for l, (h, w) in enumerate(levels) if h * w for st in range((h + 3) // 4) for x in range(w)]
        assert seq == want
        assert len(t) == greedy(levels, nb)
    # the bench workload: 46 levels of a 640x480 frame, 64 frames
    from oracle import oracle
    lr, lc, _ = oracle.pyramid_plan(480, 640, 4, 10)
    levels = [tuple(int(v) for v in oracle.hog_dims(int(r), int(c), 4)) for r, c in zip(lr, lc)]
    cells = sum(h * w for h, w in levels)
    assert cells == 140725
    n64 = len(tiles_of(levels, 64))
    assert 568.5 <= n64 / 64 <= 569.5 and n64 / 64 < 597, n64 / 64
