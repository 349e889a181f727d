"""Multi-GPU sharding of the detection path: one process per GPU, frames sharded, no data-path
collective except ONE gather of candidate lists per batch.

The path shards embarrassingly (frames are independent; the model, ~0.5 MB, is replicated), so the only
exchange is the variable-length candidate list.  Each rank contributes a fixed-capacity record
`[count | count x stride int32 words | padding]`; `torch.distributed.all_gather` (RCCL over xGMI on the
GPU box, gloo in the CPU tests) moves it.  The payload is KBs-MBs, i.e. latency-bound.
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
    """[count | records] int32 payload of fixed size 1 + cap*stride; frame ids are made global."""
    m = min(int(n), cap)
    out = np.zeros(1 + cap * stride, np.int32)
    out[0] = m
    if m:
        rec = buf[: m * stride].reshape(m, stride).copy()
        rec[:, 0] += frame_offset
        out[1:1 + m * stride] = rec.ravel()
    return out


def unpack_gathered(payloads: List[np.ndarray], stride: int) -> np.ndarray:
    """Concatenate the records of every rank (rank order = frame order for contiguous shards)."""
    parts = []
    for p in payloads:
        m = int(p[0])
        parts.append(p[1:1 + m * stride].reshape(m, stride))
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, stride), np.int32)


class CandidateGatherer:
    """The per-batch gather with its buffers allocated once: a pinned host staging buffer for the payload, one
    device tensor for the send side and one [world, payload] tensor for the receive side (a single
    ``all_gather_into_tensor``).  ``root_only`` lets the other ranks skip the copy back and the unpacking (the
    collective itself is still the one all_gather every rank takes part in)."""

    def __init__(self, stride: int, cap: int, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.stride, self.cap, self.device = stride, cap, torch.device(device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.n = 1 + cap * stride
        pin = self.device.type == "cuda"
        self.host_send = torch.zeros(self.n, dtype=torch.int32, pin_memory=pin)
        self.host_recv = torch.zeros(self.world * self.n, dtype=torch.int32, pin_memory=pin)
        self.dev_send = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        self.dev_recv = torch.zeros(self.world * self.n, dtype=torch.int32, device=self.device)

    def gather(self, buf: np.ndarray, n: int, frame_offset: int, root_only: bool = False):
        if int(n) > self.cap:
            raise RuntimeError(f"{n} candidates exceed the gather capacity {self.cap}")
        m, stride = int(n), self.stride
        hs = self.host_send.numpy()
        hs[0] = m
        if m:
            rec = hs[1:1 + m * stride].reshape(m, stride)
            rec[:] = buf[: m * stride].reshape(m, stride)
            rec[:, 0] += frame_offset
        if self.world == 1:
            return hs[1:1 + m * stride].reshape(m, stride).copy()
        # only the used prefix crosses PCIe; the collective moves the fixed-size payload
        self.dev_send[: 1 + m * stride].copy_(self.host_send[: 1 + m * stride], non_blocking=True)
        self.dist.all_gather_into_tensor(self.dev_recv, self.dev_send)
        if root_only and self.rank != 0:
            return None
        counts = self.dev_recv[:: self.n].cpu().numpy()          # world ints
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


def gather_candidates(buf: np.ndarray, n: int, stride: int, cap: int, frame_offset: int, device) -> np.ndarray:
    """One all_gather of the candidate payloads; every rank returns the concatenated (N, stride) records."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    payload = pack_candidates(buf, n, stride, cap, frame_offset)
    if world == 1:
        return unpack_gathered([payload], stride)
    send = torch.from_numpy(payload).to(device)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    return unpack_gathered([r.cpu().numpy() for r in recv], stride)
