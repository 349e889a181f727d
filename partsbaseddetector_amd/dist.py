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


class CandidateGatherer:
    """The per-batch gather with its buffers allocated once: a pinned host staging buffer for the payload, one
    device tensor for the send side and one [world, payload] tensor for the receive side (a single
    ``all_gather_into_tensor``).  ``root_only`` lets the other ranks skip the copy back and the unpacking of the
    records (every rank still takes part in the collective and reads the `world` counts, which is what keeps the
    grow-on-overflow decision identical on all ranks).

    ``cap`` is the initial capacity in records; size it from the detector's ``max_candidates`` when memory allows
    -- a too-small value costs one extra collective the first time it overflows, never an error."""

    def __init__(self, stride: int, cap: int, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.stride, self.device = stride, torch.device(device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.collectives = 0          # all_gather calls issued so far (tests / bench: 1 per step in steady state)
        self.grown = 0                # how often the buffers had to grow
        self._copy_done = None        # event after the last H2D copy out of host_send (cuda only)
        self._pending = None          # (work handle, true count, records beyond the capacity) between begin() and finish()
        self._alloc(max(int(cap), 1))

    def _alloc(self, cap: int):
        torch = self.torch
        self.cap = cap
        self.n = 1 + cap * self.stride
        pin = self.device.type == "cuda"
        self.host_send = torch.zeros(self.n, dtype=torch.int32, pin_memory=pin)
        self.host_recv = torch.zeros(self.world * self.n, dtype=torch.int32, pin_memory=pin)
        self.dev_send = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        self.dev_recv = torch.zeros(self.world * self.n, dtype=torch.int32, device=self.device)

    def _wait_host_send_free(self):
        # the previous step's asynchronous copy out of the pinned buffer must have finished before it is rewritten
        if self._copy_done is not None:
            self._copy_done.synchronize()
            self._copy_done = None

    # ---- the gather in two halves, so that a caller can put the next batch's compute between them -------------------
    # begin(): pack + H2D + the collective, issued asynchronously (`async_op=True`: RCCL runs it on its own stream next
    # to the detection kernels, gloo on its worker threads); finish(): read the counts, grow-and-repeat on overflow,
    # copy the records out.  One collective per batch, every rank issues them in the same order, and a finish() always
    # precedes the next begin(), so the buffers are never rewritten under a collective in flight.
    def begin(self, buf: np.ndarray, n: int, frame_offset: int):
        if self._pending is not None:
            raise RuntimeError("CandidateGatherer.begin(): the previous gather was not finished")
        n, stride = int(n), self.stride
        self._wait_host_send_free()
        m = min(n, self.cap)
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
        work = None
        if self.world > 1:
            # only the used prefix crosses PCIe; the collective moves the fixed-size payload
            self.dev_send[: 1 + m * stride].copy_(self.host_send[: 1 + m * stride], non_blocking=True)
            if self.device.type == "cuda":
                self._copy_done = self.torch.cuda.Event()
                self._copy_done.record(self.torch.cuda.current_stream(self.device))
            work = self.dist.all_gather_into_tensor(self.dev_recv, self.dev_send, async_op=True)
            self.collectives += 1
        self._pending = (work, n, spill)

    @property
    def pending(self) -> bool:
        return self._pending is not None

    def finish(self, root_only: bool = False):
        if self._pending is None:
            raise RuntimeError("CandidateGatherer.finish(): no gather in flight")
        work, n, spill = self._pending
        self._pending = None
        stride = self.stride
        hs = self.host_send.numpy()
        if self.world == 1:
            counts = np.array([n])
        else:
            work.wait()
            counts = self.dev_recv[:: self.n].cpu().numpy()          # world ints, read by EVERY rank
        need = int(counts.max())
        if need > self.cap:
            # some rank did not fit: every rank sees that, grows to the same capacity and repeats the collective with its
            # own records (global frame ids already applied) -- synchronously, this is the rare path
            m = min(n, self.cap)
            mine = hs[1:1 + m * stride].reshape(m, stride).copy()
            if spill is not None:
                mine = np.concatenate([mine, spill], axis=0)
            self._wait_host_send_free()
            self._alloc(_grown(need))
            self.grown += 1
            self.begin(mine.ravel(), n, 0)
            return self.finish(root_only)
        if self.world == 1:
            return hs[1:1 + n * stride].reshape(n, stride).copy()
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


def detect_level_sharded(det, im, gatherer: "CandidateGatherer", root_only: bool = False):
    """ONE frame over all ranks (secondary partitioning): every rank holds the frame, computes its
    share of the pyramid levels (`pbd_set_level_shard`: longest-processing-time assignment over the level sizes) and the
    candidate lists are gathered with the usual single collective; the records come back sorted by
    (frame, level, component, y, x) like a single-GPU call."""
    det.hd.set_level_shard(gatherer.rank, gatherer.world)
    cands = det.detect(im)
    stride = det.hd.stride
    buf = np.zeros(max(len(cands), 1) * stride, np.int32)
    for i, c in enumerate(cands):
        r = buf[i * stride:(i + 1) * stride]
        r[0], r[1], r[2], r[3], r[4] = c.frame, c.component, c.level, c.root[0], c.root[1]
        r[5:6] = np.float32(c.score()).view(np.int32)
        r[6] = len(c.parts)
        r[8:8 + 4 * len(c.parts)] = c.parts.ravel()
    rec = gatherer.gather(buf, len(cands), frame_offset=0, root_only=root_only)
    if rec is None:
        return None
    order = np.lexsort((rec[:, 3], rec[:, 4], rec[:, 1], rec[:, 2], rec[:, 0]))
    return rec[order]
