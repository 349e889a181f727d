"""world_size-2 gloo rehearsal of the multi-GPU path: frames sharded in contiguous blocks, ONE
all_gather of fixed-capacity candidate payloads, concatenation in rank (= frame) order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from partsbaseddetector_amd import dist as pd


def test_shard_range_covers_all_frames():
    for n in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            got = [pd.shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            for a, b in zip(got, got[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in got]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip_and_overflow():
    stride = 8 + 4 * 3
    rng = np.random.default_rng(0)
    buf = rng.integers(0, 1000, 5 * stride).astype(np.int32)
    p = pd.pack_candidates(buf, 5, stride, cap=8, frame_offset=10)
    assert p.size == 1 + 8 * stride and p[0] == 5
    rec = pd.unpack_gathered([p], stride)
    assert rec.shape == (5, stride)
    assert np.array_equal(rec[:, 1:], buf.reshape(5, stride)[:, 1:])
    assert np.array_equal(rec[:, 0], buf.reshape(5, stride)[:, 0] + 10)
    short = pd.pack_candidates(buf, 5, stride, cap=3)               # does not fit: the true count stays visible ...
    assert short[0] == 5
    with pytest.raises(OverflowError):                              # ... and unpacking refuses instead of truncating
        pd.unpack_gathered([short], stride)
    assert pd.unpack_gathered([pd.pack_candidates(buf, 0, stride, cap=4)], stride).shape == (0, stride)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nframes, stride, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = pd.shard_range(nframes, rank, world)
    # fake per-rank detections: frame-local ids 0.., 1 + (global frame % 3) records per frame
    recs = []
    for f in range(b, e):
        for j in range(1 + f % 3):
            r = np.zeros(stride, np.int32)
            r[0] = f - b            # local frame index, as pbd_detect_batch returns it
            r[1:] = 1000 * f + j
            recs.append(r)
    buf = np.concatenate(recs) if recs else np.zeros(0, np.int32)
    got = pd.gather_candidates(buf, len(recs), stride, cap=64, frame_offset=b, device="cpu")
    dist.barrier()
    if rank == 0:
        out.put(got)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nframes", [(2, 7), (2, 1)])
def test_gather_candidates_world2(world, nframes):
    stride = 8 + 4 * 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, stride, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = []
    for f in range(nframes):
        for j in range(1 + f % 3):
            want.append([f] + [1000 * f + j] * (stride - 1))
    assert np.array_equal(got, np.asarray(want, np.int32).reshape(-1, stride))


def _gatherer_worker(rank, world, port, q):
    import torch.distributed as dist
    from partsbaseddetector_amd import dist as pd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        stride = 12
        g = pd.CandidateGatherer(stride, cap=32, device="cpu")
        out = []
        for step in range(3):                          # buffers are reused across steps
            n = 3 + rank + step
            buf = (np.arange(n * stride, dtype=np.int32) + 1000 * rank + 7 * step).copy()
            buf.reshape(n, stride)[:, 0] = np.arange(n)
            rec = g.gather(buf, n, frame_offset=100 * rank, root_only=(step == 2))
            out.append(None if rec is None else rec.copy())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_candidate_gatherer_world2():
    """CandidateGatherer (what bench.py uses for N > 1): same records as the one-shot gather, every step,
    root_only leaves the other rank without a copy."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    stride = 12
    for step in range(3):
        want = []
        for rank in range(2):
            n = 3 + rank + step
            b = (np.arange(n * stride, dtype=np.int32) + 1000 * rank + 7 * step).reshape(n, stride).copy()
            b[:, 0] = np.arange(n) + 100 * rank
            want.append(b)
        want = np.concatenate(want)
        assert np.array_equal(res[0][step], want)
        if step == 2:
            assert res[1][step] is None
        else:
            assert np.array_equal(res[1][step], want)


def _overflow_worker(rank, world, port, q):
    import torch.distributed as dist
    from partsbaseddetector_amd import dist as pd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        stride = 8 + 4 * 26                            # the person model's record
        g = pd.CandidateGatherer(stride, cap=8192, device="cpu")
        log = []
        # step 0: both fit; step 1: rank 1 holds 30 000 records (the 8 x 1080p batch of round 1 produced 25 635) while
        # rank 0 holds 5; step 2: back to small counts on the grown buffers; step 3: rank 0 empty
        for step, counts in enumerate([(40, 50), (5, 30000), (7, 3), (0, 9)]):
            n = counts[rank]
            buf = np.zeros(n * stride, np.int32)
            r = buf.reshape(n, stride)
            r[:, 0] = np.arange(n) % 8
            r[:, 1] = rank
            r[:, 2] = step
            r[:, 3] = np.arange(n)
            before = g.collectives
            rec = g.gather(buf, n, frame_offset=8 * rank, root_only=True)
            log.append((g.collectives - before, g.cap, None if rec is None else rec[:, :4].copy()))
        q.put((rank, log))
    finally:
        dist.destroy_process_group()


def test_gatherer_overflow_grows_instead_of_raising():
    """A rank with more candidates than the gather capacity must not raise before the collective (the other ranks
    would block inside it): every rank enters, all see the true counts, all grow, and the collective is repeated."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(2):
        ncoll = [e[0] for e in res[rank]]
        caps = [e[1] for e in res[rank]]
        assert ncoll == [1, 2, 1, 1], ncoll           # one collective per step; the overflow step repeats it once
        assert caps[0] == 8192 and caps[1] >= 30000 and caps[1] == caps[2] == caps[3]
        assert all(e[2] is None for e in res[1])       # root_only
    for step, counts in enumerate([(40, 50), (5, 30000), (7, 3), (0, 9)]):
        rec = res[0][step][2]
        assert rec.shape[0] == sum(counts)
        want_rank = np.concatenate([np.full(c, r) for r, c in enumerate(counts)])
        assert np.array_equal(rec[:, 1], want_rank) and np.all(rec[:, 2] == step)
        want_idx = np.concatenate([np.arange(c) for c in counts])
        assert np.array_equal(rec[:, 3], want_idx)
        assert np.array_equal(rec[:, 0], want_idx % 8 + 8 * want_rank)      # frame ids made global


def _pipelined_worker(rank, world, port, q):
    import torch.distributed as dist
    from partsbaseddetector_amd import dist as pd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        stride = 12
        g = pd.CandidateGatherer(stride, cap=16, device="cpu")
        out = []
        buf = np.zeros(64 * stride, np.int32)           # ONE buffer, rewritten every step like the detector's
        # the bench's order: detect(k) ; finish(k-1) ; begin(k).  Step 2 overflows on rank 1 (40 > 16 records).
        counts = [(3, 5), (7, 2), (4, 40), (6, 6)]
        for step, c in enumerate(counts):
            n = c[rank]
            r = buf[: n * stride].reshape(n, stride)
            r[:] = 1000 * step + 100 * rank
            r[:, 0] = np.arange(n)
            r[:, 1] = np.arange(n)
            if g.pending:
                out.append(g.finish(root_only=True))
            g.begin(buf, n, frame_offset=50 * rank)
            buf[:] = -1                                   # the caller's buffer is free after begin()
        out.append(g.finish(root_only=True))
        assert not g.pending
        q.put((rank, out, g.collectives, g.grown))
    finally:
        dist.destroy_process_group()


def test_gatherer_begin_finish_pipelined_with_overflow():
    """begin() / finish() one step apart (the collective of batch k runs under the compute of batch k+1): same records
    as the synchronous form, the caller's buffer may be rewritten right after begin(), an overflow is repaired in the
    finish() that discovers it, and every rank issues the same number of collectives."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, out, ncoll, grown = q.get(timeout=180)
        res[rank] = (out, ncoll, grown)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    stride = 12
    counts = [(3, 5), (7, 2), (4, 40), (6, 6)]
    assert res[0][1] == res[1][1] == 5 and res[0][2] == res[1][2] == 1      # 4 steps + one repeat; grown once
    assert all(o is None for o in res[1][0])                                   # root_only
    for step, c in enumerate(counts):
        rec = res[0][0][step]
        assert rec.shape == (sum(c), stride)
        want_frame = np.concatenate([np.arange(c[0]), np.arange(c[1]) + 50])
        assert np.array_equal(rec[:, 0], want_frame)
        assert np.array_equal(rec[:, 1], np.concatenate([np.arange(c[0]), np.arange(c[1])]))
        assert np.array_equal(rec[:, 2], np.concatenate([np.full(c[0], 1000 * step), np.full(c[1], 1000 * step + 100)]))


def _device_path_worker(rank, world, port, q):
    """the device-resident path of the gatherer (what bench.py uses for N > 1), rehearsed with CPU tensors: the "kernels"
    are numpy writes into g.payload (a tensor of the FULL capacity, of which the collective sends a prefix)"""
    import torch.distributed as dist
    from partsbaseddetector_amd import dist as pd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        stride = 12
        g = pd.CandidateGatherer(stride, cap=16, device="cpu", cap_full=64)
        counts = [(3, 5), (7, 2), (4, 40), (6, 6), (0, 1)]

        def emit(payload, cap, step):
            n = counts[step][rank]
            p = payload.numpy()
            p[0] = n                                          # the TRUE count, also when it does not fit
            m = min(n, cap)
            rec = p[1:1 + m * stride].reshape(m, stride)
            rec[:] = 1000 * step + 100 * rank
            rec[:, 0] = np.arange(m) + 50 * rank              # global frame ids, written by the "walk kernel"
            rec[:, 1] = np.arange(m)

        out, payloads = [], []
        for step in range(len(counts)):
            # the order of dist.DeviceBatchGather.submit: the "kernels" of batch k fill this batch's payload tensor first,
            # then the gather of batch k-1 is finished (an overflow is repaired from ITS tensor, which is still intact)
            payloads.append(g.payload.data_ptr())
            emit(g.payload, g.cap_full, step)
            if g.pending:
                out.append(g.finish(root_only=True))
            g.begin_device(None)
        out.append(g.finish(root_only=True))
        try:                                                   # a list longer than the payload tensors: every rank raises
            emit(g.payload, g.cap_full, 2)
            g.payload.numpy()[0] = 1000
            g.begin_device(None)
            g.finish()
            too_long = "no error"
        except OverflowError as e:
            too_long = str(e)
        q.put((rank, out, g.collectives, g.grown, payloads[0] != payloads[1], too_long))
    finally:
        dist.destroy_process_group()


def test_gatherer_device_path_pipelined_with_overflow():
    """begin_device(): no host packing at all -- a prefix of the payload tensor is handed to the collective as it is;
    consecutive batches use alternating payload tensors; an overflow makes every rank grow and send a longer prefix of the
    same tensor (nothing is recomputed, the next batch may already be in flight)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_device_path_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, out, ncoll, grown, alternates, too_long = q.get(timeout=180)
        res[rank] = (out, ncoll, grown, alternates)
        assert "raise max_candidates" in too_long
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    stride = 12
    counts = [(3, 5), (7, 2), (4, 40), (6, 6), (0, 1)]
    assert res[0][1] == res[1][1] == 7 and res[0][2] == res[1][2] == 1      # 5 steps + one repeat + the too-long list; grown once
    assert res[0][3] and res[1][3]
    assert all(o is None for o in res[1][0])
    for step, c in enumerate(counts):
        rec = res[0][0][step]
        assert rec.shape == (sum(c), stride)
        assert np.array_equal(rec[:, 0], np.concatenate([np.arange(c[0]), np.arange(c[1]) + 50]))
        assert np.array_equal(rec[:, 1], np.concatenate([np.arange(c[0]), np.arange(c[1])]))
        assert np.array_equal(rec[:, 2], np.concatenate([np.full(c[0], 1000 * step), np.full(c[1], 1000 * step + 100)]))


def _rank_failed_worker(rank, world, port, q):
    import torch.distributed as dist
    from partsbaseddetector_amd import dist as pd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        stride = 12
        g = pd.CandidateGatherer(stride, cap=16, device="cpu")
        buf = np.zeros(4 * stride, np.int32)
        try:
            g.gather(buf, -1 if rank == 1 else 4, frame_offset=0)          # rank 1's detect step failed
            q.put((rank, "no error"))
        except pd.RankFailed as e:
            q.put((rank, str(e)))
        # the group is still usable: nobody is stuck inside the collective
        rec = g.gather(buf, 2, frame_offset=0)
        q.put((rank, rec.shape))
    finally:
        dist.destroy_process_group()


def test_a_failed_rank_does_not_block_the_others():
    """a rank whose detect raised used to skip the collective and leave the others blocked in all_gather.  Now
    it enters with a negative count and every rank raises RankFailed after the collective."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_failed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    msgs = {r: m for r, m in got if isinstance(m, str)}
    assert set(msgs) == {0, 1} and all("rank(s) [1]" in m for m in msgs.values())
    assert sorted(m for r, m in got if not isinstance(m, str)) == [(4, 12), (4, 12)]


def _lanes_worker(rank, world, port, q):
    """bench.py's multi-stream loop over RCCL, rehearsed with CPU tensors: K lanes, each with its own CandidateGatherer; step s
    goes to lane s % K, whose submit finishes the lane's previous gather (step s - K) and issues step s's collective"""
    import torch.distributed as dist
    from partsbaseddetector_amd import dist as pd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        stride, K, steps = 12, 3, 8
        lanes = [pd.CandidateGatherer(stride, cap=8, device="cpu", cap_full=64) for _ in range(K)]

        def emit(g, step):
            n = 1 + (step * 5 + rank * 3) % 11             # step 4 on rank 1 overflows the initial capacity of 8
            p = g.payload.numpy()
            p[0] = n
            rec = p[1:1 + n * stride].reshape(n, stride)
            rec[:] = 1000 * step + 100 * rank
            rec[:, 0] = np.arange(n) + 50 * rank
            return n

        out, sent = [], []
        for s in range(steps):
            g = lanes[s % K]
            sent.append(emit(g, s))
            if g.pending:
                out.append(g.finish(root_only=True))        # the records of step s - K
            g.begin_device(None)
        m = min(K, steps)
        for i in range(m):                                   # drain in submission order
            out.append(lanes[(steps - m + i) % K].finish(root_only=True))
        q.put((rank, out, sent, sum(g.collectives for g in lanes), sum(g.grown for g in lanes)))
    finally:
        dist.destroy_process_group()


def test_several_gatherers_in_flight_round_robin():
    """K lanes with a gatherer each (bench.py --streams K at N > 1): up to K collectives are in flight, every rank issues them
    in the same order, the records come back per step in submission order, and an overflow on one lane grows that lane only."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lanes_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, out, sent, ncoll, grown = q.get(timeout=180)
        res[rank] = (out, sent, ncoll, grown)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    stride, steps = 12, 8
    assert all(o is None for o in res[1][0]) and len(res[0][0]) == steps
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] >= 1          # the same collectives (incl. repeats) on both ranks
    for s in range(steps):
        rec = res[0][0][s]
        n0, n1 = res[0][1][s], res[1][1][s]
        assert rec.shape == (n0 + n1, stride)
        assert np.array_equal(rec[:, 0], np.concatenate([np.arange(n0), np.arange(n1) + 50]))
        assert np.array_equal(rec[:, 2], np.concatenate([np.full(n0, 1000 * s), np.full(n1, 1000 * s + 100)]))
