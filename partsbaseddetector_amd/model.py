"""Model container mirroring the reference ``Model`` (include/Model.hpp:49-122) and the
``Parts`` index tables (include/Parts.hpp:172-187), plus seeded synthetic model builders.

The reference's model files (Person_26parts.xml, Face_68parts.xml) are not in its tree
(``models/`` is an un-vendored submodule), so tests and bench use synthetic models with the same
geometry.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import synth


@dataclass
class Model:
    """Field names follow the reference (Model.hpp): filtersw_, biasw_, anchors_, defw_, and the
    per-component index tables filterid_/biasid_/defid_/parentid_."""

    name: str = "synthetic"
    interval: int = 10          # Model::nscales_ is really the interval (src/FileStorageModel.cpp:105)
    thresh: float = 0.0
    sbin: int = 4
    norient: int = 18
    flen: int = 32
    filtersw: List[np.ndarray] = field(default_factory=list)  # each (k, k*flen) float64, channel fastest
    biasw: List[float] = field(default_factory=list)
    anchors: List[tuple] = field(default_factory=list)         # (x, y), 0-based
    defw: List[List[float]] = field(default_factory=list)      # 4 floats each
    filterid: List[List[List[int]]] = field(default_factory=list)  # [c][p][mix]
    biasid: List[List[List[int]]] = field(default_factory=list)    # [c][p][...]
    defid: List[List[List[int]]] = field(default_factory=list)     # [c][p][mix] (root: [])
    parentid: List[List[int]] = field(default_factory=list)        # [c][p], root -1

    # ---- accessors named as in the reference -------------------------------------------------
    def ncomponents(self) -> int:
        return len(self.filterid)

    def nparts(self, c: int = 0) -> int:
        return len(self.filterid[c])

    def max_parts(self) -> int:
        return max(len(x) for x in self.filterid)

    def nfilters(self) -> int:
        return len(self.filtersw)

    def validate(self) -> None:
        """Structural checks before the tables reach pbd_create / the oracle; raises ValueError (never `assert`, which
        `python -O` strips)."""
        def need(cond, msg):
            if not cond:
                raise ValueError(f"model {self.name!r}: {msg}")
        need(self.norient == 18, "the reference's uu/vv tables hold 9 orientations (src/HOGFeatures.cpp:192-193): norient must be 18")
        need(self.flen >= self.norient + self.norient // 2 + 5, f"flen {self.flen} too small for norient {self.norient}")
        need(self.ncomponents() >= 1 and len(self.filtersw) >= 1, "no components / filters")
        need(len(self.biasid) == len(self.defid) == len(self.parentid) == self.ncomponents(), "indexer tables differ in length")
        for f in self.filtersw:
            f = np.asarray(f)
            need(f.ndim == 2 and f.shape[0] >= 1 and f.shape == (f.shape[0], f.shape[0] * self.flen), f"filter shape {f.shape}")
        for c in range(self.ncomponents()):
            np_c = self.nparts(c)
            need(np_c >= 1 and len(self.biasid[c]) == len(self.defid[c]) == len(self.parentid[c]) == np_c,
                 f"component {c}: indexer tables differ in length")
            for p in range(np_c):
                par = self.parentid[c][p]
                need((par == -1 and p == 0) or (p > 0 and 0 <= par < p), f"component {c} part {p}: parent {par} breaks the topological order")
                need(len(self.filterid[c][p]) >= 1, f"component {c} part {p} has no mixtures")
                for f in self.filterid[c][p]:
                    need(0 <= f < len(self.filtersw), f"component {c} part {p}: filter id {f} out of range")
                if p > 0:
                    K = len(self.filterid[c][p])
                    L = len(self.filterid[c][par])
                    need(len(self.defid[c][p]) >= K and len(self.biasid[c][p]) >= K,
                         f"component {c} part {p}: {K} mixtures but {len(self.defid[c][p])} defids / {len(self.biasid[c][p])} biasids")
                    for mm in range(K):
                        need(0 <= self.defid[c][p][mm] < len(self.defw), f"component {c} part {p}: defid {self.defid[c][p][mm]} out of range")
                        need(len(self.anchors) >= len(self.defw), "fewer anchors than deformations")
                        need(0 <= self.biasid[c][p][mm] and self.biasid[c][p][mm] + L <= len(self.biasw),
                             f"component {c} part {p}: biasid {self.biasid[c][p][mm]} out of range")
                else:
                    need(len(self.biasid[c][p]) >= 1 and 0 <= self.biasid[c][p][0] < len(self.biasw), f"component {c}: root biasid out of range")

    def flatten(self) -> "FlatModel":
        return FlatModel(self)


class FlatModel:
    """Plain arrays in the layout both C interfaces take (include/pbd.h ``pbd_model`` and the
    oracle's ``pbdo_model`` share it field for field)."""

    def __init__(self, m: Model):
        m.validate()
        self.model = m
        self.ncomponents = m.ncomponents()
        self.nfilters = m.nfilters()
        self.flen = m.flen
        self.filter_ksize = np.array([f.shape[0] for f in m.filtersw], dtype=np.int32)
        sizes = np.array([f.size for f in m.filtersw], dtype=np.int64)
        self.filter_offset = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.filters_f64 = np.concatenate([np.asarray(f, dtype=np.float64).ravel() for f in m.filtersw])
        # distributeModel converts the filters to T (src/PartsBasedDetector.cpp:114-117)
        self.filters_f32 = self.filters_f64.astype(np.float32)
        self.biasw = np.asarray(m.biasw, dtype=np.float32)
        self.defw = np.asarray(m.defw, dtype=np.float32).reshape(-1, 4)
        self.anchors = np.asarray(m.anchors, dtype=np.int32).reshape(-1, 2)
        part_offset, parentid, mix_offset, filterid, biasid, defid = [0], [], [0], [], [], []
        for c in range(self.ncomponents):
            for p in range(m.nparts(c)):
                parentid.append(m.parentid[c][p])
                K = len(m.filterid[c][p])
                for mm in range(K):
                    filterid.append(m.filterid[c][p][mm])
                    bid = m.biasid[c][p]
                    biasid.append(bid[mm] if mm < len(bid) else -1)
                    did = m.defid[c][p]
                    defid.append(did[mm] if (p > 0 and mm < len(did)) else -1)
                mix_offset.append(mix_offset[-1] + K)
            part_offset.append(part_offset[-1] + m.nparts(c))
        self.part_offset = np.asarray(part_offset, dtype=np.int32)
        self.parentid = np.asarray(parentid, dtype=np.int32)
        self.mix_offset = np.asarray(mix_offset, dtype=np.int32)
        self.filterid = np.asarray(filterid, dtype=np.int32)
        self.biasid = np.asarray(biasid, dtype=np.int32)
        self.defid = np.asarray(defid, dtype=np.int32)
        self.thresh = float(np.float32(m.thresh))
        self.sbin, self.interval, self.norient = m.sbin, m.interval, m.norient
        self.max_parts = m.max_parts()
        # back-pointer slots: (part gp, parent mixture m) -> ptr_slot[gp] + m
        slots, total = [], 0
        for c in range(self.ncomponents):
            p0 = part_offset[c]
            for p in range(m.nparts(c)):
                slots.append(total)
                par = m.parentid[c][p]
                if par >= 0:
                    total += len(m.filterid[c][par])
        self.ptr_slot = np.asarray(slots, dtype=np.int32)
        self.nslots = total


def synthetic_model(seed: int = 26, pa=None, nmix: int = 6, ncomponents: int = 1, ksize=5,
                    sbin: int = 4, interval: int = 10, thresh: float = 0.0, linear_def: bool = False,
                    anchor_range: int = 4, filter_sigma: float = 0.05, bias_sigma: float = 0.1,
                    share_filters: bool = False, name: str = "synthetic") -> Model:
    """Seeded model with the layout the Matlab builder produces (matlab/learning/buildmodel.m:27-75):
    per part K filters, K deformations ([0.01 0 0.01 0] unless ``linear_def``), and an L x K bias
    table allocated child-major so that ``bias(mm)[m] = biasw[base + mm*L + m]``.

    pa: 1-based parent table (0 for the root).  Components get independent parameter sets unless
    ``share_filters`` (face-model style: components index one shared filter pool)."""
    if pa is None:
        pa = synth.PERSON_PA
    nparts = len(pa)
    flen, norient = 32, 18
    m = Model(name=name, interval=interval, thresh=thresh, sbin=sbin, norient=norient, flen=flen)
    stream = 100

    def draw(n):
        nonlocal stream
        stream += 1
        return synth.normalish(seed, n, stream)

    def draw_int(n, lo, hi):
        nonlocal stream
        stream += 1
        return synth.randint(seed, n, lo, hi, stream)

    shared_fid = None
    for c in range(ncomponents):
        fid_c, bid_c, did_c, par_c = [], [], [], []
        for p in range(nparts):
            parent = pa[p] - 1
            par_c.append(parent)
            K = nmix
            # filters
            if share_filters and shared_fid is not None:
                fid = shared_fid[p]
            else:
                fid = []
                for _ in range(K):
                    # ksize may be a list: filter sizes then cycle through it (a model with filters of several sizes)
                    ks = ksize[len(m.filtersw) % len(ksize)] if isinstance(ksize, (list, tuple)) else ksize
                    w = draw(ks * ks * flen) * filter_sigma
                    # a few exact zeros exercise the skipped-tap rule (src/filter.cpp:3818-3856)
                    w[draw_int(8, 0, w.size - 1)] = 0.0
                    fid.append(len(m.filtersw))
                    m.filtersw.append(w.reshape(ks, ks * flen))
            fid_c.append(fid)
            # bias
            if parent < 0:
                bid_c.append([len(m.biasw)])
                m.biasw.append(float(draw(1)[0] * bias_sigma))
                did_c.append([])
            else:
                L = len(fid_c[parent])
                base = len(m.biasw)
                m.biasw.extend((draw(K * L) * bias_sigma).tolist())
                # row-major flattening of the L x K table; row 0 = base + mm*L
                bid_c.append([base + mm * L + l for l in range(L) for mm in range(K)])
                dids = []
                ax = draw_int(K, -anchor_range, anchor_range)
                ay = draw_int(K, -anchor_range, anchor_range)
                lin = draw(2 * K) * 0.01
                for mm in range(K):
                    dids.append(len(m.defw))
                    if linear_def:
                        m.defw.append([0.01 + 0.002 * mm, float(lin[2 * mm]), 0.012 + 0.001 * mm, float(lin[2 * mm + 1])])
                    else:
                        m.defw.append([0.01, 0.0, 0.01, 0.0])
                    m.anchors.append((int(ax[mm]), int(ay[mm])))
                did_c.append(dids)
        if share_filters and shared_fid is None:
            shared_fid = fid_c
        m.filterid.append(fid_c)
        m.biasid.append(bid_c)
        m.defid.append(did_c)
        m.parentid.append(par_c)
    m.validate()
    return m


# threshold chosen so that synthetic "scene" frames give O(10-100) candidates per 640x480 frame
# (measured with the oracle on seeds 1..4; see tests/golden/make_golden.py)
PERSON_THRESH = 18.85


def synthetic_person_model(thresh: float | None = None) -> Model:
    """26 parts x 6 mixtures = 156 filters of 5x5x32, sbin 4, interval 10."""
    return synthetic_model(seed=26, pa=synth.PERSON_PA, nmix=6, thresh=PERSON_THRESH if thresh is None else thresh,
                           name="synthetic_person_26parts")


def synthetic_face_model(thresh: float = 0.0, nparts: int = 20, ncomponents: int = 3, interval: int = 5) -> Model:
    """Zhu-Ramanan-style stand-in for Face_68parts: many parts, 1 mixture per part, several
    components over one shared filter pool (BASELINE.json configs[0])."""
    pa = [0] + [max(1, i - (i % 3)) for i in range(1, nparts)]
    return synthetic_model(seed=68, pa=pa, nmix=1, ncomponents=ncomponents, interval=interval, thresh=thresh,
                           share_filters=True, name="synthetic_face")


def synthetic_tiny_model(thresh: float = 0.0, linear_def: bool = True) -> Model:
    """3 parts x 2 mixtures, used by the small committed fixtures."""
    return synthetic_model(seed=3, pa=[0, 1, 1], nmix=2, thresh=thresh, linear_def=linear_def, interval=5,
                           name="synthetic_tiny")
