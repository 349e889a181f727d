"""Host-side mirror of the reference's operator interface for the detection hot path, over the
C ABI (include/pbd.h).  Same names and argument meaning as the reference:

    IFeatures / HOGFeatures<T>            include/IFeatures.hpp:49-73, src/HOGFeatures.cpp
    IConvolutionEngine / Spatial...       include/IConvolutionEngine.hpp:44-68, src/SpatialConvolutionEngine.cpp
    DynamicProgram<T>                     include/DynamicProgram.hpp:74-75, src/DynamicProgram.cpp
    PartsBasedDetector<T>                 include/PartsBasedDetector.hpp:152-175, src/PartsBasedDetector.cpp
    Candidate                             include/Candidate.hpp:56-99

Every compute call runs HIP kernels through libpbd_hip.so; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import PbdError
from .model import FlatModel, Model


@dataclass
class Candidate:
    """include/Candidate.hpp:56-80: part rectangles (x, y, w, h), confidences, component.
    `frame`, `level`, `root` record where the candidate was back-tracked from."""

    parts: np.ndarray
    confidence: np.ndarray
    component: int
    frame: int = 0
    level: int = 0
    root: tuple = (0, 0)

    def score(self) -> float:  # Candidate.hpp:82
        return float(self.confidence[0]) if len(self.confidence) else float("-inf")

    @staticmethod
    def sort(candidates: List["Candidate"]) -> None:  # Candidate.hpp:91-99 (descending by score)
        candidates.sort(key=lambda c: -c.score())

    def boundingBox(self):
        """Candidate.hpp:105-111: hull of the part rectangles (cv::Rect operator|), as (x, y, w, h)."""
        x, y, w, h = (int(v) for v in self.parts[0])
        for r in self.parts:
            bx, by, bw, bh = (int(v) for v in r)
            if w <= 0 or h <= 0:            # a.empty(): a = b
                x, y, w, h = bx, by, bw, bh
            elif bw > 0 and bh > 0:
                x1, y1 = min(x, bx), min(y, by)
                w, h = max(x + w, bx + bw) - x1, max(y + h, by + bh) - y1
                x, y = x1, y1
        return x, y, w, h

    @staticmethod
    def nonMaximaSuppression(im_shape, candidates: List["Candidate"], overlap: float = 0.0) -> None:
        """Candidate.hpp:277-304: greedy paint-the-canvas suppression on the bounding boxes, in the given
        order (callers sort by score first: cells/detect.cpp:237-238, ros/Node.cpp:192-196).  In place."""
        rows, cols = int(im_shape[0]), int(im_shape[1])
        scratch = np.zeros((rows, cols), np.uint8)
        keep = 0
        for cand in list(candidates):
            x, y, w, h = cand.boundingBox()
            x1, y1 = max(x, 0), max(y, 0)                         # box & bounds (cv::Rect operator&)
            x2, y2 = min(x + w, cols), min(y + h, rows)
            if x2 - x1 <= 0 or y2 - y1 <= 0:
                x1 = y1 = x2 = y2 = 0
            area = (x2 - x1) * (y2 - y1)
            boxsum = float(scratch[y1:y2, x1:x2].sum())
            ratio = boxsum / area if area else float("nan")       # NaN > overlap is false: an empty box is kept
            if ratio > overlap:
                continue
            scratch[y1:y2, x1:x2] = 1
            candidates[keep] = cand
            keep += 1
        del candidates[keep:]


class Handle:
    """Owns a pbd_handle (one handle = one host thread = one GPU)."""

    def __init__(self, model, device: int = 0, conv_mode: int = _lib.CONV_EXACT, max_batch: int = 1,
                 max_candidates: int = 1 << 18, stream: Optional[int] = None, real_type: int = _lib.REAL_F32):
        self.lib = _lib.load()
        self.flat: FlatModel = model if isinstance(model, FlatModel) else model.flatten()
        self._cm = _lib.c_model(self.flat)
        cfg = _lib.CConfig(device, real_type, conv_mode, max_batch, max_candidates, stream)
        h = C.c_void_p()
        rc = self.lib.pbd_create(C.byref(self._cm), C.byref(cfg), C.byref(h))
        if rc != _lib.PBD_OK:
            raise PbdError(rc, self.lib.pbd_last_error(None).decode())
        self.h = h
        self.dtype = np.float32 if real_type == _lib.REAL_F32 else np.float64   # reference template parameter T
        self.max_batch = max_batch
        self.max_candidates = max_candidates
        self.stride = self.lib.pbd_candidate_stride(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.pbd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, allow=()):
        if rc != _lib.PBD_OK and rc not in allow:
            raise PbdError(rc, self.lib.pbd_last_error(self.h).decode())
        return rc

    def stream_ptr(self) -> int:
        """pbd_stream: the hipStream_t the handle's kernels run on (wrap it with torch.cuda.ExternalStream to order torch work behind it)"""
        return int(self.lib.pbd_stream(self.h) or 0)

    def set_level_shard(self, rank: int, world: int) -> None:
        """pbd_set_level_shard: this handle computes only its share of the pyramid levels of each frame"""
        self.check(self.lib.pbd_set_level_shard(self.h, rank, world))

    # ---- helpers -------------------------------------------------------------------------------
    def plan(self, rows: int, cols: int):
        n = C.c_int()
        arrs = [np.zeros(_lib.MAX_LEVELS, np.int32) for _ in range(4)]
        sc = np.zeros(_lib.MAX_LEVELS, np.float32)
        self.check(self.lib.pbd_pyramid_plan(self.h, rows, cols, C.byref(n), *[_lib.ptr(a, C.c_int) for a in arrs],
                                             _lib.ptr(sc, C.c_float)))
        k = n.value
        return {"nlevels": k, "img_rows": arrs[0][:k].copy(), "img_cols": arrs[1][:k].copy(),
                "feat_rows": arrs[2][:k].copy(), "feat_cols": arrs[3][:k].copy(), "scales": sc[:k].copy()}

    def unpack_candidates(self, buf: np.ndarray, n: int) -> List[Candidate]:
        out = []
        rec = buf[: n * self.stride].reshape(n, self.stride)
        for r in rec:
            npart = int(r[6])
            conf = np.zeros(npart, np.float32)
            conf[0] = r[5:6].view(np.float32)[0]
            out.append(Candidate(parts=r[8:8 + 4 * npart].reshape(npart, 4).copy(), confidence=conf,
                                 component=int(r[1]), frame=int(r[0]), level=int(r[2]), root=(int(r[3]), int(r[4]))))
        return out

    def profile(self, on=True):
        """on: False / 0 off, True / 1 every kernel, 2 the convolution only (pbd_profile_enable)"""
        self.check(self.lib.pbd_profile_enable(self.h, int(on)))
        self.check(self.lib.pbd_profile_reset(self.h))

    def profile_read(self):
        out = {}
        for k, name in enumerate(_lib.KERNELS):
            ms, n = C.c_double(), C.c_int()
            self.check(self.lib.pbd_profile_read(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def get_stage(self, stage: int, frame: int, level: int, rows: int, cols: int):
        planes = {_lib.STAGE_FEATURES: None, _lib.STAGE_RESPONSES: self.flat.nfilters,
                  _lib.STAGE_ROOTV: self.flat.ncomponents, _lib.STAGE_ROOTI: self.flat.ncomponents}[stage]
        if stage == _lib.STAGE_FEATURES:
            dst = np.empty((rows, cols * self.flat.flen), self.dtype)
        elif stage == _lib.STAGE_ROOTI:
            dst = np.empty((planes, rows, cols), np.int32)
        else:
            dst = np.empty((planes, rows, cols), self.dtype)
        self.check(self.lib.pbd_get_stage(self.h, stage, frame, level, dst.ctypes.data, dst.nbytes))
        return dst


class HOGFeatures:
    """IFeatures (include/IFeatures.hpp:49-73) as implemented by HOGFeatures<float>."""

    def __init__(self, handle: Handle):
        self.hd = handle
        self._scales = np.zeros(0, np.float32)

    def binsize(self) -> int:
        return self.hd.lib.pbd_binsize(self.hd.h)

    def nscales(self) -> int:
        return len(self._scales)

    def scales(self) -> np.ndarray:
        return self._scales

    def pyramid(self, im: np.ndarray) -> List[np.ndarray]:
        """pyramid(im, pyrafeatures): list of (H, W*flen) maps of T, fine to coarse."""
        if im.dtype not in _lib.DEPTH_CODE:
            # src/HOGFeatures.cpp:136-146: 8U / 16U / 32F / 64F, anything else is CV_StsUnsupportedFormat
            raise PbdError(-2, f"image dtype {im.dtype}: uint8, uint16, float32 or float64")
        if im.ndim == 2:
            im = im[:, :, None]
        rows, cols, cn = im.shape
        es = im.dtype.itemsize
        if not im.flags.c_contiguous and not (im.strides[2] == es and im.strides[1] == cn * es):
            im = np.ascontiguousarray(im)
        plan = self.hd.plan(rows, cols)
        feats = [np.empty((int(r), int(c) * self.hd.flat.flen), self.hd.dtype)
                 for r, c in zip(plan["feat_rows"], plan["feat_cols"])]
        arr = _lib.ptr_array(feats)
        self.hd.check(self.hd.lib.pbd_features_pyramid(self.hd.h, im.ctypes.data, rows, cols, cn, im.strides[0],
                                                       _lib.DEPTH_CODE[im.dtype], arr))
        self._scales = plan["scales"]
        return feats

    def level_images(self, rows: int, cols: int, cn: int, dtype=np.uint8) -> List[np.ndarray]:
        """the resampled pyramid images of the last pyramid()/detect() call (frame 0), for tests"""
        plan = self.hd.plan(rows, cols)
        out = []
        for l in range(plan["nlevels"]):
            img = np.empty((int(plan["img_rows"][l]), int(plan["img_cols"][l]), cn), dtype)
            self.hd.check(self.hd.lib.pbd_get_pyramid_image(self.hd.h, 0, l, img.ctypes.data))
            out.append(img)
        return out


class SpatialConvolutionEngine:
    """IConvolutionEngine (include/IConvolutionEngine.hpp:44-68)."""

    def __init__(self, handle: Handle):
        self.hd = handle

    def setFilters(self, filters: Sequence[np.ndarray]) -> None:
        fl = [np.ascontiguousarray(f, self.hd.dtype) for f in filters]
        ks = np.array([f.shape[0] for f in fl], np.int32)
        arr = _lib.ptr_array(fl)
        self.hd.check(self.hd.lib.pbd_conv_set_filters(self.hd.h, len(fl), arr, _lib.ptr(ks, C.c_int)))
        self._nfilters = len(fl)

    def pdf(self, features: Sequence[np.ndarray]) -> List[np.ndarray]:
        """pdf(features, responses): responses[level] is (nfilters, H, W); responses[level][filter] as in the reference."""
        flen = self.hd.flat.flen
        feats = [np.ascontiguousarray(f, self.hd.dtype) for f in features]
        rows = np.array([f.shape[0] for f in feats], np.int32)
        cols = np.array([f.shape[1] // flen for f in feats], np.int32)
        nf = getattr(self, "_nfilters", self.hd.flat.nfilters)
        resp = [np.empty((nf, int(r), int(c)), self.hd.dtype) for r, c in zip(rows, cols)]
        self.hd.check(self.hd.lib.pbd_conv_pdf(self.hd.h, len(feats), _lib.ptr_array(feats), _lib.ptr(rows, C.c_int),
                                               _lib.ptr(cols, C.c_int), _lib.ptr_array(resp)))
        return resp


class DynamicProgram:
    """DynamicProgram<float> (include/DynamicProgram.hpp:74-75)."""

    def __init__(self, handle: Handle):
        self.hd = handle

    def min(self, scores: Sequence[np.ndarray]):
        """min(parts, scores, Ix, Iy, Ik, rootv, rooti); scores[level] is (nfilters, H, W).
        Returns per level: Ix, Iy, Ik as (nslots, H, W) int32 (slot = pbd_ptr_slot(c, part) + parent mixture),
        rootv (ncomponents, H, W) float32, rooti (ncomponents, H, W) int32."""
        sc = [np.ascontiguousarray(s, self.hd.dtype) for s in scores]
        rows = np.array([s.shape[1] for s in sc], np.int32)
        cols = np.array([s.shape[2] for s in sc], np.int32)
        ns, nc = max(self.hd.flat.nslots, 1), self.hd.flat.ncomponents
        Ix = [np.zeros((ns, int(r), int(c)), np.int32) for r, c in zip(rows, cols)]
        Iy = [np.zeros((ns, int(r), int(c)), np.int32) for r, c in zip(rows, cols)]
        Ik = [np.zeros((ns, int(r), int(c)), np.int32) for r, c in zip(rows, cols)]
        rootv = [np.empty((nc, int(r), int(c)), self.hd.dtype) for r, c in zip(rows, cols)]
        rooti = [np.empty((nc, int(r), int(c)), np.int32) for r, c in zip(rows, cols)]
        self.hd.check(self.hd.lib.pbd_dp_min(self.hd.h, len(sc), _lib.ptr(rows, C.c_int), _lib.ptr(cols, C.c_int),
                                             _lib.ptr_array(sc), _lib.ptr_array(Ix), _lib.ptr_array(Iy),
                                             _lib.ptr_array(Ik), _lib.ptr_array(rootv), _lib.ptr_array(rooti)))
        return Ix, Iy, Ik, rootv, rooti

    def argmin(self, scales: np.ndarray, capacity: Optional[int] = None) -> List[Candidate]:
        """argmin(parts, rootv, rooti, scales, Ix, Iy, Ik, candidates) on the result of the last min()."""
        cap = capacity or self.hd.max_candidates
        buf = np.zeros(cap * self.hd.stride, np.int32)
        n = C.c_int()
        sc = np.ascontiguousarray(scales, np.float32)
        self.hd.check(self.hd.lib.pbd_dp_argmin(self.hd.h, _lib.ptr(sc, C.c_float), buf.ctypes.data, cap, C.byref(n)))
        return self.hd.unpack_candidates(buf, n.value)


class PartsBasedDetector:
    """PartsBasedDetector<float> (include/PartsBasedDetector.hpp:152-175)."""

    def __init__(self, device: int = 0, conv_mode: int = _lib.CONV_EXACT, max_batch: int = 1,
                 max_candidates: int = 1 << 18, stream: Optional[int] = None, dtype=np.float32):
        """dtype: the reference's template parameter T (float32 as src/demo.cpp:85, float64 as the ECTO/ROS callers)."""
        self._kw = dict(device=device, conv_mode=conv_mode, max_batch=max_batch, max_candidates=max_candidates,
                        stream=stream, real_type=_lib.REAL_F32 if np.dtype(dtype) == np.float32 else _lib.REAL_F64)
        self.hd: Optional[Handle] = None
        self._name = ""

    def name(self) -> str:
        return self._name

    def distributeModel(self, model: Model) -> None:
        """src/PartsBasedDetector.cpp:102-127: creates the feature / convolution engines and the DP."""
        if self.hd is not None:
            self.hd.close()
        self.hd = Handle(model, **self._kw)
        self._name = getattr(model, "name", "")
        self.features_ = HOGFeatures(self.hd)
        self.convolution_engine_ = SpatialConvolutionEngine(self.hd)
        self.dp_ = DynamicProgram(self.hd)

    def _need(self):
        if self.hd is None:
            raise PbdError(-5, "detect() before distributeModel()")

    def detect(self, im: np.ndarray, depth: Optional[np.ndarray] = None, capacity: Optional[int] = None) -> List[Candidate]:
        """detect(im[, depth], candidates); `depth` is ignored exactly as in the reference (:91-93)."""
        self._need()
        if im.dtype not in _lib.DEPTH_CODE:
            raise PbdError(-2, f"image dtype {im.dtype}: uint8, uint16, float32 or float64 (src/HOGFeatures.cpp:136-146)")
        if im.ndim == 2:
            im = im[:, :, None]
        es = im.dtype.itemsize
        if not (im.strides[2] == es and im.strides[1] == im.shape[2] * es):
            im = np.ascontiguousarray(im)
        rows, cols, cn = im.shape
        cap = capacity or self.hd.max_candidates
        buf = np.zeros(cap * self.hd.stride, np.int32)
        n = C.c_int()
        if im.dtype == np.uint8:
            self.hd.check(self.hd.lib.pbd_detect(self.hd.h, im.ctypes.data, rows, cols, cn, im.strides[0], buf.ctypes.data,
                                                 cap, C.byref(n)))
        else:
            self.hd.check(self.hd.lib.pbd_detect_typed(self.hd.h, im.ctypes.data, rows, cols, cn, im.strides[0],
                                                       _lib.DEPTH_CODE[im.dtype], buf.ctypes.data, cap, C.byref(n)))
        self.features_._scales = self.hd.plan(rows, cols)["scales"]
        return self.hd.unpack_candidates(buf, n.value)

    def detect_batch(self, frames: Sequence[np.ndarray], capacity: Optional[int] = None) -> List[Candidate]:
        self._need()
        fr = [np.ascontiguousarray(f if f.ndim == 3 else f[:, :, None], np.uint8) for f in frames]
        rows, cols, cn = fr[0].shape
        assert all(f.shape == fr[0].shape for f in fr), "a batch holds equally sized frames"
        cap = capacity or self.hd.max_candidates
        buf = np.zeros(cap * self.hd.stride, np.int32)
        n = C.c_int()
        self.hd.check(self.hd.lib.pbd_detect_batch(self.hd.h, len(fr), _lib.ptr_array(fr), rows, cols, cn, cols * cn,
                                                   buf.ctypes.data, cap, C.byref(n)))
        return self.hd.unpack_candidates(buf, n.value)

    def submit_batch(self, frames: Sequence[np.ndarray]) -> None:
        """pbd_detect_batch_submit: stage + transfer + enqueue the whole path for `frames` without waiting (at most
        two batches in flight); `wait_batch` returns the results in submission order."""
        self._need()
        fr = [np.ascontiguousarray(f if f.ndim == 3 else f[:, :, None], np.uint8) for f in frames]
        rows, cols, cn = fr[0].shape
        assert all(f.shape == fr[0].shape for f in fr), "a batch holds equally sized frames"
        self.hd.check(self.hd.lib.pbd_detect_batch_submit(self.hd.h, len(fr), _lib.ptr_array(fr), rows, cols, cn, cols * cn))

    def wait_batch(self, capacity: Optional[int] = None, raw: bool = False):
        self._need()
        cap = capacity or self.hd.max_candidates
        if not hasattr(self, "_buf") or self._buf.size < cap * self.hd.stride:
            self._buf = np.zeros(cap * self.hd.stride, np.int32)
        n = C.c_int()
        self.hd.check(self.hd.lib.pbd_detect_batch_wait(self.hd.h, self._buf.ctypes.data, cap, C.byref(n)))
        if raw:
            return self._buf, n.value
        return self.hd.unpack_candidates(self._buf, n.value)

    def submit_batch_device(self, d_frames_ptr: int, nframes: int, rows: int, cols: int, cn: int) -> None:
        """pbd_detect_batch_device_submit: like submit_batch for frames already resident in device memory"""
        self._need()
        self.hd.check(self.hd.lib.pbd_detect_batch_device_submit(self.hd.h, nframes, d_frames_ptr, rows, cols, cn))

    def detect_batch_device_out(self, d_frames_ptr: int, nframes: int, rows: int, cols: int, cn: int, frame_offset: int,
                                d_payload_ptr: int, capacity: int) -> None:
        """pbd_detect_batch_device_out: the whole path, candidate list left on the device in int32[1 + capacity*stride]
        ([found | sorted records], frame ids + frame_offset); asynchronous on the handle's stream"""
        self._need()
        self.hd.check(self.hd.lib.pbd_detect_batch_device_out(self.hd.h, nframes, d_frames_ptr, rows, cols, cn, frame_offset,
                                                              d_payload_ptr, capacity))

    def argmin_device_out(self, frame_offset: int, d_payload_ptr: int, capacity: int) -> None:
        """pbd_argmin_device_out: re-emit the candidate list of the batch still resident on the device (after an overflow)"""
        self._need()
        self.hd.check(self.hd.lib.pbd_argmin_device_out(self.hd.h, frame_offset, d_payload_ptr, capacity))

    def detect_batch_device(self, d_frames_ptr: int, nframes: int, rows: int, cols: int, cn: int,
                            capacity: Optional[int] = None, raw: bool = False):
        """frames already resident in device memory (e.g. a torch uint8 tensor's data_ptr())."""
        self._need()
        cap = capacity or self.hd.max_candidates
        if not hasattr(self, "_buf") or self._buf.size < cap * self.hd.stride:
            self._buf = np.zeros(cap * self.hd.stride, np.int32)
        n = C.c_int()
        self.hd.check(self.hd.lib.pbd_detect_batch_device(self.hd.h, nframes, d_frames_ptr, rows, cols, cn,
                                                          self._buf.ctypes.data, cap, C.byref(n)))
        if raw:
            return self._buf, n.value
        return self.hd.unpack_candidates(self._buf, n.value)


class DetectorPool:
    """K detectors of one model on one GPU -- K handles, i.e. K HIP streams and K workspaces -- fed round-robin.

    A handle runs its kernels in order on its own stream; the kernels of a small batch do not fill the chip (a single frame's
    launches are smaller than the chip).  Batches submitted to DIFFERENT handles overlap at kernel granularity.  Measured on
    MI355X: one 640x480 frame per step 423 -> 525 detections/s with four handles, one 1920x1080 frame 83 -> 103 with three;
    batches that fill the chip (64 x 640x480, 8 x 1920x1080) gain nothing (profiles/r03_bench_*s*.json,
    profiles/r03_streams_*.txt).  Results come back in submission order; every batch is computed by exactly one handle, so
    they are those of a single PartsBasedDetector.

        pool = DetectorPool(model, n=3, max_batch=64)
        for batch in stream:                       # frames resident on the device
            pool.submit_batch_device(ptr, nframes, rows, cols, cn)
            while pool.ready_before_next_submit: handle(pool.wait_batch())
        while pool.pending: handle(pool.wait_batch())
    """

    def __init__(self, model: Model, n: int = 3, **kw):
        if n < 1:
            raise ValueError("DetectorPool needs at least one detector")
        self.dets: List[PartsBasedDetector] = []
        try:
            for _ in range(n):
                d = PartsBasedDetector(**kw)
                d.distributeModel(model)
                self.dets.append(d)
        except Exception:           # a later handle failed (e.g. out of memory): release the earlier ones
            self.close()
            raise
        self._submitted = 0         # batches submitted so far
        self._collected = 0         # batches handed back so far

    @property
    def pending(self) -> int:
        return self._submitted - self._collected

    @property
    def ready_before_next_submit(self) -> bool:
        """True when the lane the next submit would use still holds an uncollected batch (collect one first)"""
        return self.pending >= len(self.dets)

    def _lane(self, i: int) -> "PartsBasedDetector":
        return self.dets[i % len(self.dets)]

    def submit_batch_device(self, d_frames_ptr: int, nframes: int, rows: int, cols: int, cn: int) -> None:
        if self.ready_before_next_submit:
            raise PbdError(-5, "DetectorPool: every lane holds an uncollected batch; call wait_batch() first")
        self._lane(self._submitted).submit_batch_device(d_frames_ptr, nframes, rows, cols, cn)
        self._submitted += 1

    def submit_batch(self, frames: Sequence[np.ndarray]) -> None:
        if self.ready_before_next_submit:
            raise PbdError(-5, "DetectorPool: every lane holds an uncollected batch; call wait_batch() first")
        self._lane(self._submitted).submit_batch(frames)
        self._submitted += 1

    def wait_batch(self, capacity: Optional[int] = None, raw: bool = False):
        """the oldest uncollected batch's candidates (as PartsBasedDetector.wait_batch)"""
        if not self.pending:
            raise PbdError(-5, "DetectorPool.wait_batch(): nothing submitted")
        out = self._lane(self._collected).wait_batch(capacity, raw)
        self._collected += 1
        return out

    def close(self) -> None:
        for d in self.dets:
            if d.hd is not None:
                d.hd.close()
