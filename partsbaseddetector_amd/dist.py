"""Multi-GPU sharding of the detection path: one process per GPU, frames sharded, no data-path
collective except ONE gather of candidate lists per batch.

The path shards embarrassingly (frames are independent; the model, ~0.5 MB, is replicated), so the only
exchange is the variable-length candidate list.  Each rank contributes a fixed-size payload
`[count | records | padding]`; `torch.distributed.all_gather_into_tensor` (RCCL over xGMI on the GPU box,
gloo in the CPU tests) moves it.  The payload is KBs-MBs, i.e. latency-bound.

Capacity never makes a rank raise before the collective (a rank that raised while the others are inside
the collective would leave them blocked until the RCCL timeout): word 0 carries the rank's TRUE count,
every rank enters the collective with the records that fit, every rank reads all counts afterwards, and
if any count exceeds the capacity all ranks grow their buffers to the same new capacity and repeat the
collective.  In steady state that is exactly one collective per batch.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_range(nframes: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of frames owned by `rank`: [begin, end).  Sizes differ by at most one."""
    base, rem = divmod(nframes, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pack_candidates(buf: np.ndarray, n: int, stride: int, cap: int, frame_offset: int = 0) -> np.ndarray:
    """[count | records] int32 payload of fixed size 1 + cap*stride; frame ids are made global.
    Word 0 is the true count `n`; only min(n, cap) records fit (`unpack_gathered` reports that as an overflow)."""
    n = int(n)
    m = min(n, cap)
    out = np.zeros(1 + cap * stride, np.int32)
    out[0] = n
    if m:
        rec = buf[: m * stride].reshape(m, stride).copy()
        rec[:, 0] += frame_offset
        out[1:1 + m * stride] = rec.ravel()
    return out


def unpack_gathered(payloads: List[np.ndarray], stride: int) -> np.ndarray:
    """Concatenate the records of every rank (rank order = frame order for contiguous shards).
    Raises OverflowError if a payload's count exceeds what the payload can hold (never truncates silently)."""
    parts = []
    for r, p in enumerate(payloads):
        m, cap = int(p[0]), (len(p) - 1) // stride
        if m > cap:
            raise OverflowError(f"rank {r} found {m} candidates, payload capacity {cap}")
        parts.append(p[1:1 + m * stride].reshape(m, stride))
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, stride), np.int32)


def _grown(need: int) -> int:
    """capacity (records) after an overflow: next power of two with 25 % head-room -- every rank computes the same value"""
    cap = 1024
    while cap < need + need // 4:
        cap *= 2
    return cap


class RankFailed(RuntimeError):
    """raised on EVERY rank, after the collective, when some rank's detect step failed (its payload carries a negative count)"""


class CandidateGatherer:
    """The per-batch gather with its buffers allocated once: one [world, payload] device tensor for the receive side (a
    single ``all_gather_into_tensor``), two payload tensors for the send side and pinned host mirrors.  Two ways in:

    * ``begin_device()``: the payload was written ON THE DEVICE by ``pbd_detect_batch_device_out`` straight into
      ``self.payload`` (count in word 0, records sorted, global frame ids) -- nothing passes through the host before the
      collective, and afterwards only ``world`` counts (every rank) and the records (rank 0) come back.  This is the
      multi-GPU path.
    * ``begin(buf, n, frame_offset)``: host records (gloo tests, level sharding): packed into a pinned buffer, copied up.

    ``root_only`` lets the other ranks skip the copy back and the unpacking of the records (every rank still takes part
    in the collective and reads the `world` counts, which is what keeps the grow-on-overflow decision identical on all
    ranks).  ``force_collective`` issues the collective at world size 1 as well: the one-GPU rehearsal of the RCCL path.

    ``cap`` is the initial capacity in records of what the COLLECTIVE moves per rank; a too-small value costs one extra
    collective the first time it overflows, never an error.  ``cap_full`` (device path) is the size of the payload tensors
    the kernels write into -- the detector's ``max_candidates``: the collective sends a prefix of that tensor, so after
    an overflow the same, still intact, tensor is simply sent again with a longer prefix (the detector may already be
    working on the next batch: nothing has to be recomputed).  Only a list longer than ``cap_full`` is an error
    (OverflowError on every rank, like PBD_ERR_CAPACITY on one GPU)."""

    def __init__(self, stride: int, cap: int, device, force_collective: bool = False, cap_full: int = 0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.stride, self.device = stride, torch.device(device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.collectives = 0          # all_gather calls issued so far (tests / bench: 1 per step in steady state)
        self.grown = 0                # how often the buffers had to grow
        self._copy_done = None        # event after the last H2D copy out of host_send (cuda only)
        self._pending = None          # state between begin*() and finish()
        self.cap_full = max(int(cap_full), int(cap), 1)
        self.dev_send = None
        self._cur = 0
        self._alloc(max(int(cap), 1))

    def _alloc(self, cap: int):
        torch = self.torch
        self.cap = cap
        self.n = 1 + cap * self.stride
        pin = self.device.type == "cuda"
        self.host_send = torch.zeros(self.n, dtype=torch.int32, pin_memory=pin)
        self.host_recv = torch.zeros(self.world * self.n, dtype=torch.int32, pin_memory=pin)
        self.dev_recv = torch.zeros(self.world * self.n, dtype=torch.int32, device=self.device)
        # two send buffers of the FULL size: the collective of batch k may still read one while batch k+1's kernels fill
        # the other; a buffer is reused only after the finish() of the batch that used it.  They survive a growth of the
        # collective's capacity (the overflowed list is in one of them) unless the host path outgrows them.
        if self.dev_send is None or cap > self.cap_full:
            self.cap_full = max(self.cap_full, cap)
            self.dev_send = [torch.zeros(1 + self.cap_full * self.stride, dtype=torch.int32, device=self.device) for _ in range(2)]

    @property
    def payload(self):
        """the device tensor int32[1 + cap_full*stride] the NEXT batch's candidate list is to be written into
        (``pbd_detect_batch_device_out(..., payload.data_ptr(), cap_full)``) before ``begin_device()``"""
        return self.dev_send[self._cur]

    def _wait_host_send_free(self):
        # the previous step's asynchronous copy out of the pinned buffer must have finished before it is rewritten
        if self._copy_done is not None:
            self._copy_done.synchronize()
            self._copy_done = None

    def _issue(self, send):
        work = None
        if self.collective:
            work = self.dist.all_gather_into_tensor(self.dev_recv, send[: self.n], async_op=True)   # a prefix: contiguous
            self.collectives += 1
        return work

    # ---- the gather in two halves, so that a caller can put the next batch's compute between them -------------------
    # begin*(): the collective, issued asynchronously (`async_op=True`: RCCL runs it on its own stream next to the
    # detection kernels, gloo on its worker threads); finish(): read the counts, grow-and-repeat on overflow, copy the
    # records out.  One collective per batch, every rank issues them in the same order, and a finish() always precedes
    # the next begin*(), so the buffers are never rewritten under a collective in flight.
    def begin_device(self, producer_stream=None):
        """`self.payload` is being filled by kernels enqueued on `producer_stream` (a torch stream, e.g.
        ``torch.cuda.ExternalStream(pbd_stream(h))``; None: torch's current stream).  Nothing is copied: the collective is
        ordered behind those kernels and reads the tensor where it is."""
        if self._pending is not None:
            raise RuntimeError("CandidateGatherer.begin_device(): the previous gather was not finished")
        torch = self.torch
        send = self.dev_send[self._cur]
        self._cur ^= 1
        if self.device.type == "cuda" and producer_stream is not None:
            ev = torch.cuda.Event()
            ev.record(producer_stream)
            torch.cuda.current_stream(self.device).wait_event(ev)      # the collective (and finish()'s reads) wait for the kernels
        self._pending = ("dev", self._issue(send), send, None, None)

    def begin(self, buf: np.ndarray, n: int, frame_offset: int):
        """host records.  n < 0 flags a failed detect step on this rank: the rank still enters the collective and every rank
        raises RankFailed from finish()."""
        if self._pending is not None:
            raise RuntimeError("CandidateGatherer.begin(): the previous gather was not finished")
        n, stride = int(n), self.stride
        self._wait_host_send_free()
        m = min(max(n, 0), self.cap)
        hs = self.host_send.numpy()
        hs[0] = n                                         # the TRUE count, also when it does not fit
        if m:
            rec = hs[1:1 + m * stride].reshape(m, stride)
            rec[:] = buf[: m * stride].reshape(m, stride)
            rec[:, 0] += frame_offset
        # the caller's buffer is free again after this call; records beyond the capacity are kept for the repeat
        spill = None
        if n > m:
            spill = np.array(buf[m * stride: n * stride], np.int32).reshape(n - m, stride)
            spill[:, 0] += frame_offset
        send = self.dev_send[self._cur]
        self._cur ^= 1
        work = None
        if self.collective:
            # only the used prefix crosses PCIe; the collective moves the fixed-size payload
            send[: 1 + m * stride].copy_(self.host_send[: 1 + m * stride], non_blocking=True)
            if self.device.type == "cuda":
                self._copy_done = self.torch.cuda.Event()
                self._copy_done.record(self.torch.cuda.current_stream(self.device))
            work = self._issue(send)
        self._pending = ("host", work, send, n, spill)

    @property
    def pending(self) -> bool:
        return self._pending is not None

    def finish(self, root_only: bool = False):
        if self._pending is None:
            raise RuntimeError("CandidateGatherer.finish(): no gather in flight")
        kind, work, send, n, spill = self._pending
        self._pending = None
        stride = self.stride
        if work is not None:
            work.wait()
            counts = self.dev_recv[:: self.n].cpu().numpy()          # world ints, read by EVERY rank
        elif kind == "dev":
            counts = send[:1].cpu().numpy()
        else:
            counts = np.array([n])
        if int(counts.min()) < 0:
            raise RankFailed(f"detect failed on rank(s) {[r for r, c in enumerate(counts) if c < 0]}")
        need = int(counts.max())
        if need > self.cap:
            # some rank did not fit: every rank sees that, grows to the same capacity and repeats the collective with its
            # own records -- synchronously, this is the rare path
            if kind == "dev" and need > self.cap_full:
                raise OverflowError(f"{need} candidates on some rank, payload tensors hold {self.cap_full}: raise max_candidates")
            self.grown += 1
            if kind == "dev":
                # the overflowed list is complete in `send` (sized cap_full, not rewritten before this finish()): send a longer prefix
                self._alloc(min(_grown(need), self.cap_full))
                self._pending = ("dev", self._issue(send), send, None, None)
                return self.finish(root_only)
            hs = self.host_send.numpy()
            m = min(n, self.cap)
            mine = hs[1:1 + m * stride].reshape(m, stride).copy()
            if spill is not None:
                mine = np.concatenate([mine, spill], axis=0)
            self._wait_host_send_free()
            self._alloc(_grown(need))
            self.begin(mine.ravel(), n, 0)
            return self.finish(root_only)
        if work is None:
            if kind == "dev":
                k = int(counts[0])
                return send[1:1 + k * stride].cpu().numpy().reshape(k, stride)
            return self.host_send.numpy()[1:1 + n * stride].reshape(n, stride).copy()
        if root_only and self.rank != 0:
            return None
        parts = []
        for r in range(self.world):
            k = int(counts[r]) * stride
            dst = self.host_recv[r * self.n + 1: r * self.n + 1 + k]
            dst.copy_(self.dev_recv[r * self.n + 1: r * self.n + 1 + k], non_blocking=True)
            parts.append(dst)
        if self.device.type == "cuda":
            self.torch.cuda.current_stream(self.device).synchronize()
        if not parts:
            return np.zeros((0, stride), np.int32)
        return np.concatenate([p.numpy().reshape(-1, stride) for p in parts], axis=0)

    def gather(self, buf: np.ndarray, n: int, frame_offset: int, root_only: bool = False):
        """begin() + finish(): the synchronous form."""
        self.begin(buf, n, frame_offset)
        return self.finish(root_only)


def gather_candidates(buf: np.ndarray, n: int, stride: int, cap: int, frame_offset: int, device) -> np.ndarray:
    """One-shot form: every rank returns the concatenated (N, stride) records.  `cap` is only the initial
    capacity (see CandidateGatherer)."""
    return CandidateGatherer(stride, cap, device).gather(buf, n, frame_offset)


class DeviceBatchGather:
    """Frames sharded over the ranks, candidate lists gathered without a host round trip: glue between a
    ``detector.PartsBasedDetector`` and a ``CandidateGatherer`` on a cuda device.

        g = DeviceBatchGather(det, CandidateGatherer(stride, cap, "cuda:0", cap_full=det.hd.max_candidates))
        for every batch k:
            records_of_k_minus_1 = g.submit(d_frames_ptr, nframes, rows, cols, cn, frame_offset, root_only=True)
        records_of_last = g.collect(root_only=True)

    ``submit`` enqueues the whole path of batch k (the candidate list goes into this batch's payload tensor, on the
    device), THEN finishes the gather of batch k-1 -- whose collective ran under batch k-1's successor being enqueued and is
    long complete -- and issues batch k's collective behind its kernels.  The detection kernels run on the detector's own
    stream; the collective is ordered behind them with an event."""

    def __init__(self, det, gatherer: CandidateGatherer):
        import torch
        self.det, self.g = det, gatherer
        self.stream = torch.cuda.ExternalStream(det.hd.stream_ptr(), device=gatherer.device)

    def submit(self, d_frames_ptr: int, nframes: int, rows: int, cols: int, cn: int, frame_offset: int, root_only: bool = False):
        pay = self.g.payload                     # not the tensor a pending gather is reading: the buffers alternate
        self.det.detect_batch_device_out(d_frames_ptr, nframes, rows, cols, cn, frame_offset, pay.data_ptr(), self.g.cap_full)
        prev = self.g.finish(root_only) if self.g.pending else None
        self.g.begin_device(self.stream)
        return prev

    def collect(self, root_only: bool = False):
        return self.g.finish(root_only) if self.g.pending else None


def detect_level_sharded(det, im, gatherer: "CandidateGatherer", root_only: bool = False):
    """ONE frame over all ranks (secondary partitioning): every rank holds the frame, computes its
    share of the pyramid levels (`pbd_set_level_shard`: longest-processing-time assignment over the level sizes) and the
    candidate lists are gathered with the usual single collective; the records come back sorted by
    (frame, level, component, y, x) like a single-GPU call.

    The shard is handle state: it is restored to (0, 1) before this returns, so a later detect() / detect_batch() on the
    same detector sees every level again.  A rank whose share fails (e.g. PBD_ERR_CAPACITY -- every rank owns different
    levels, so only one may fail) still enters the collective, flagged, and ALL ranks raise RankFailed afterwards: nobody
    is left blocked in the all_gather."""
    from ._lib import PbdError
    stride = det.hd.stride
    cands, err = [], None
    det.hd.set_level_shard(gatherer.rank, gatherer.world)
    try:
        cands = det.detect(im)
    except PbdError as e:
        err = e
    finally:
        det.hd.set_level_shard(0, 1)
    buf = np.zeros(max(len(cands), 1) * stride, np.int32)
    for i, c in enumerate(cands):
        r = buf[i * stride:(i + 1) * stride]
        r[0], r[1], r[2], r[3], r[4] = c.frame, c.component, c.level, c.root[0], c.root[1]
        r[5:6] = np.float32(c.score()).view(np.int32)
        r[6] = len(c.parts)
        r[8:8 + 4 * len(c.parts)] = c.parts.ravel()
    try:
        rec = gatherer.gather(buf, -1 if err is not None else len(cands), frame_offset=0, root_only=root_only)
    except RankFailed as rf:
        raise RankFailed(f"{rf}; this rank: {err}") from err
    if rec is None:
        return None
    order = np.lexsort((rec[:, 3], rec[:, 4], rec[:, 1], rec[:, 2], rec[:, 0]))
    return rec[order]
