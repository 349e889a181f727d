"""ctypes wrapper around the CPU oracle (oracle/libpbd_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  PARITY UNPINNED (see pbd_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAX_LEVELS = 128


def build(force: bool = False) -> str:
    # PBD_ORACLE_SO: another build of the same sources (the sanitised one, a -O3 one) -- tests/test_oracle_sanitized.py and
    # bench.py's cpu_baseline use it; the file must already exist
    alt = os.environ.get("PBD_ORACLE_SO")
    if alt:
        if not os.path.exists(alt):
            raise FileNotFoundError(alt)
        return alt
    so = os.path.join(_HERE, "libpbd_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pbd_oracle.c", "pbd_oracle_impl.inc", "pbd_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


class _Model(C.Structure):
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


class _Hdr(C.Structure):
    _fields_ = [("component", C.c_int), ("level", C.c_int), ("root_x", C.c_int), ("root_y", C.c_int),
                ("score", C.c_float), ("nparts", C.c_int)]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.pbdo_detect_f32.restype = C.c_int
        _LIB.pbdo_detect_f64.restype = C.c_int
        _LIB.pbdo_detect_d_f32.restype = C.c_int
        _LIB.pbdo_detect_d_f64.restype = C.c_int
    return _LIB


def c_model(flat):
    """flat: partsbaseddetector_amd.model.FlatModel.  Keeps references alive on the struct."""
    m = _Model()
    m.ncomponents, m.nfilters, m.flen = flat.ncomponents, flat.nfilters, flat.flen
    m.filter_ksize = _p(flat.filter_ksize, C.c_int)
    m.filter_offset = _p(flat.filter_offset, C.c_int64)
    m.filters_f32 = _p(flat.filters_f32, C.c_float)
    m.filters_f64 = _p(flat.filters_f64, C.c_double)
    m.nbias, m.biasw = len(flat.biasw), _p(flat.biasw, C.c_float)
    m.ndefs, m.defw, m.anchors = len(flat.defw), _p(flat.defw, C.c_float), _p(flat.anchors, C.c_int)
    m.part_offset, m.parentid = _p(flat.part_offset, C.c_int), _p(flat.parentid, C.c_int)
    m.mix_offset, m.filterid = _p(flat.mix_offset, C.c_int), _p(flat.filterid, C.c_int)
    m.biasid, m.defid = _p(flat.biasid, C.c_int), _p(flat.defid, C.c_int)
    m.thresh, m.sbin, m.interval, m.norient = flat.thresh, flat.sbin, flat.interval, flat.norient
    m._keep = flat
    return m


def num_threads() -> int:
    return lib().pbdo_num_threads()


def set_num_threads(n: int) -> None:
    lib().pbdo_set_num_threads(int(n))


DEPTH_CODE = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 2, np.dtype(np.float32): 5, np.dtype(np.float64): 6}   # OpenCV's depth codes


def _image(im):
    """contiguous (rows, cols, cn) image of one of the depths HOGFeatures::pyramid accepts, and its depth code"""
    im = np.asarray(im)
    if im.dtype not in DEPTH_CODE:
        im = im.astype(np.uint8)
    im = np.ascontiguousarray(im)
    if im.ndim == 2:
        im = im[:, :, None]
    return im, DEPTH_CODE[im.dtype]


def pyramid_plan(rows, cols, sbin, interval):
    lr = np.zeros(MAX_LEVELS, np.int32)
    lc = np.zeros(MAX_LEVELS, np.int32)
    sc = np.zeros(MAX_LEVELS, np.float32)
    n = lib().pbdo_pyramid_plan(rows, cols, sbin, interval, _p(lr, C.c_int), _p(lc, C.c_int), _p(sc, C.c_float))
    if n <= 0:
        raise ValueError(f"pyramid_plan failed ({n}) for {rows}x{cols}")
    return lr[:n].copy(), lc[:n].copy(), sc[:n].copy()


def hog_dims(rows, cols, sbin):
    a, b = C.c_int(), C.c_int()
    lib().pbdo_hog_dims(rows, cols, sbin, C.byref(a), C.byref(b))
    return a.value, b.value


def resize_linear_u8(src, drows, dcols):
    src = np.ascontiguousarray(src, np.uint8)
    if src.ndim == 2:
        src = src[:, :, None]
    r, c, cn = src.shape
    dst = np.empty((drows, dcols, cn), np.uint8)
    lib().pbdo_resize_linear_u8(_p(src, C.c_uint8), r, c, cn, C.c_size_t(c * cn), _p(dst, C.c_uint8), drows, dcols,
                                C.c_size_t(dcols * cn))
    return dst


def pyrdown_u8(src):
    src = np.ascontiguousarray(src, np.uint8)
    if src.ndim == 2:
        src = src[:, :, None]
    r, c, cn = src.shape
    dst = np.empty(((r + 1) // 2, (c + 1) // 2, cn), np.uint8)
    lib().pbdo_pyrdown_u8(_p(src, C.c_uint8), r, c, cn, C.c_size_t(c * cn), _p(dst, C.c_uint8), C.c_size_t(dst.shape[1] * cn))
    return dst


def pyramid_images(im, sbin, interval):
    im, depth = _image(im)
    r, c, cn = im.shape
    lr, lc, sc = pyramid_plan(r, c, sbin, interval)
    n = len(lr)
    tot = int(sum(int(a) * int(b) * cn for a, b in zip(lr, lc)))
    out = np.empty(tot, im.dtype)
    off = np.zeros(n + 1, np.int64)
    lr2, lc2, sc2 = np.zeros(MAX_LEVELS, np.int32), np.zeros(MAX_LEVELS, np.int32), np.zeros(MAX_LEVELS, np.float32)
    lib().pbdo_pyramid_images(C.c_void_p(im.ctypes.data), depth, r, c, cn, C.c_size_t(c * cn), sbin, interval,
                              C.c_void_p(out.ctypes.data), _p(off, C.c_int64), _p(lr2, C.c_int), _p(lc2, C.c_int), _p(sc2, C.c_float))
    imgs = [out[off[l]:off[l + 1]].reshape(lr[l], lc[l], cn) for l in range(n)]
    return imgs, sc


def hog_features(im, sbin=4, norient=18, flen=32, dtype=np.float32):
    im, depth = _image(im)
    r, c, cn = im.shape
    oh, ow = hog_dims(r, c, sbin)
    feat = np.empty((oh, ow * flen), dtype)
    fn = lib().pbdo_hog_features_d_f32 if dtype == np.float32 else lib().pbdo_hog_features_d_f64
    fn(C.c_void_p(im.ctypes.data), depth, r, c, cn, C.c_size_t(c * cn), sbin, norient, flen,
       _p(feat, C.c_float if dtype == np.float32 else C.c_double))
    return feat


def conv(feat, filt, flen=32):
    """feat (H, W*flen), filt (k, k*flen) -> (H, W) response ("same" correlation, border 0 / 1 on last channel)."""
    dtype = feat.dtype
    feat = np.ascontiguousarray(feat)
    filt = np.ascontiguousarray(filt, dtype)
    H, W = feat.shape[0], feat.shape[1] // flen
    k = filt.shape[0]
    resp = np.empty((H, W), dtype)
    if dtype == np.float32:
        lib().pbdo_conv_f32(_p(feat, C.c_float), H, W, flen, _p(filt, C.c_float), k, _p(resp, C.c_float))
    else:
        lib().pbdo_conv_f64(_p(feat, C.c_double), H, W, flen, _p(filt, C.c_double), k, _p(resp, C.c_double))
    return resp


def dt(score, ax, bx, ay, by, osx, osy):
    dtype = score.dtype
    score = np.ascontiguousarray(score)
    M, N = score.shape
    out = np.empty((M, N), dtype)
    Ix = np.empty((M, N), np.int32)
    Iy = np.empty((M, N), np.int32)
    fn = lib().pbdo_dt_f32 if dtype == np.float32 else lib().pbdo_dt_f64
    ct = C.c_float if dtype == np.float32 else C.c_double
    fn(_p(score, ct), M, N, C.c_double(ax), C.c_double(bx), C.c_double(ay), C.c_double(by), osx, osy, _p(out, ct),
       _p(Ix, C.c_int), _p(Iy, C.c_int))
    return out, Ix, Iy


def dp_min(flat, c, responses):
    """responses (nfilters, H, W) -> Ix, Iy, Ik (nslots, H, W) int32, rootv (H, W), rooti (H, W)."""
    dtype = responses.dtype
    responses = np.ascontiguousarray(responses)
    _, H, W = responses.shape
    ns = max(flat.nslots, 1)
    Ix = np.zeros((ns, H, W), np.int32)
    Iy = np.zeros((ns, H, W), np.int32)
    Ik = np.zeros((ns, H, W), np.int32)
    rootv = np.empty((H, W), dtype)
    rooti = np.empty((H, W), np.int32)
    cm = c_model(flat)
    fn = lib().pbdo_dp_min_f32 if dtype == np.float32 else lib().pbdo_dp_min_f64
    ct = C.c_float if dtype == np.float32 else C.c_double
    fn(C.byref(cm), c, _p(responses, ct), H, W, _p(Ix, C.c_int), _p(Iy, C.c_int), _p(Ik, C.c_int), _p(rootv, ct),
       _p(rooti, C.c_int))
    return Ix, Iy, Ik, rootv, rooti


def _unpack(hdr, rects, n, max_parts):
    out = []
    for i in range(n):
        h = hdr[i]
        out.append({
            "component": h.component, "level": h.level, "root_x": h.root_x, "root_y": h.root_y,
            "score": float(np.float32(h.score)), "parts": rects[i, :h.nparts].copy(),
        })
    return out


def dp_argmin(flat, c, level, scale, Ix, Iy, Ik, rootv, rooti, capacity=100000):
    dtype = rootv.dtype
    H, W = rootv.shape
    hdr = (_Hdr * capacity)()
    mp = flat.max_parts
    rects = np.zeros((capacity, mp, 4), np.int32)
    cm = c_model(flat)
    fn = lib().pbdo_dp_argmin_f32 if dtype == np.float32 else lib().pbdo_dp_argmin_f64
    ct = C.c_float if dtype == np.float32 else C.c_double
    n = fn(C.byref(cm), c, level, C.c_float(scale), H, W, _p(np.ascontiguousarray(Ix), C.c_int),
           _p(np.ascontiguousarray(Iy), C.c_int), _p(np.ascontiguousarray(Ik), C.c_int),
           _p(np.ascontiguousarray(rootv), ct), _p(np.ascontiguousarray(rooti), C.c_int), hdr, _p(rects, C.c_int), mp,
           capacity)
    if n < 0:
        raise RuntimeError("candidate capacity overflow")
    return _unpack(hdr, rects, n, mp)


def features_pyramid(flat, im, dtype=np.float32):
    """IFeatures::pyramid -> list of (H, W*flen) feature maps, scales."""
    im, depth = _image(im)
    r, c, cn = im.shape
    lr, lc, _ = pyramid_plan(r, c, flat.sbin, flat.interval)
    n = len(lr)
    dims = [hog_dims(int(a), int(b), flat.sbin) for a, b in zip(lr, lc)]
    tot = sum(h * w * flat.flen for h, w in dims)
    feat = np.empty(max(tot, 1), dtype)
    off = np.zeros(MAX_LEVELS + 1, np.int64)
    orow, ocol = np.zeros(MAX_LEVELS, np.int32), np.zeros(MAX_LEVELS, np.int32)
    sc = np.zeros(MAX_LEVELS, np.float32)
    cm = c_model(flat)
    fn = lib().pbdo_features_d_f32 if dtype == np.float32 else lib().pbdo_features_d_f64
    ct = C.c_float if dtype == np.float32 else C.c_double
    got = fn(C.byref(cm), C.c_void_p(im.ctypes.data), depth, r, c, cn, C.c_size_t(c * cn), _p(feat, ct), _p(off, C.c_int64),
             _p(orow, C.c_int), _p(ocol, C.c_int), _p(sc, C.c_float))
    assert got == n
    feats = [feat[off[l]:off[l + 1]].reshape(dims[l][0], dims[l][1] * flat.flen) for l in range(n)]
    return feats, sc[:n].copy()


def responses(flat, feat):
    """IConvolutionEngine::pdf for one level: (H, W*flen) -> (nfilters, H, W)."""
    dtype = feat.dtype
    feat = np.ascontiguousarray(feat)
    H, W = feat.shape[0], feat.shape[1] // flat.flen
    resp = np.empty((flat.nfilters, H, W), dtype)
    cm = c_model(flat)
    if dtype == np.float32:
        lib().pbdo_responses_f32(C.byref(cm), _p(feat, C.c_float), H, W, _p(resp, C.c_float))
    else:
        lib().pbdo_responses_f64(C.byref(cm), _p(feat, C.c_double), H, W, _p(resp, C.c_double))
    return resp


def detect(flat, im, dtype=np.float32, capacity=200000, want_stage_ms=False):
    """PartsBasedDetector<T>::detect -> list of candidate dicts sorted by (level, component, y, x)."""
    im, depth = _image(im)
    r, c, cn = im.shape
    hdr = (_Hdr * capacity)()
    mp = flat.max_parts
    rects = np.zeros((capacity, mp, 4), np.int32)
    ms = (C.c_double * 5)()
    cm = c_model(flat)
    fn = lib().pbdo_detect_d_f32 if dtype == np.float32 else lib().pbdo_detect_d_f64
    n = fn(C.byref(cm), C.c_void_p(im.ctypes.data), depth, r, c, cn, C.c_size_t(c * cn), hdr, _p(rects, C.c_int), mp, capacity, ms)
    if n < 0:
        raise RuntimeError(f"oracle detect failed ({n})")
    cands = _unpack(hdr, rects, n, mp)
    if want_stage_ms:
        return cands, dict(zip(["pyramid", "hog", "conv", "dp_min", "argmin"], list(ms)))
    return cands
