#!/usr/bin/env python3
"""bench.py -- detections/sec of the detection hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path (pyramid -> HOG -> filter-bank correlation -> distance
transform / dynamic program -> back-tracking) over one batch of synthetic 640x480 frames that are
already resident in HBM; candidates come back to the host, and with N > 1 every rank's candidate list
is gathered on rank 0 with one RCCL all_gather (frames are sharded, weak scaling: each rank owns its
own batch).  Workload = BASELINE.json configs[2] (person model, batch of 64 640x480 frames, full path
on one GPU).  Rank 0 prints ONE JSON line.

Convolution mode: `value` is measured with the bit-exact kernel (PBD_CONV_EXACT: every response, score and
index identical to the reference order).  At N=1 the same batch is then also timed with the matrix-core
kernel (PBD_CONV_MFMA: operands split into two bf16 terms, fp32 accumulation; responses within the
north-star tolerance of 1e-4) and reported as `fast_mode`, together with a record-by-record comparison of
its candidates with the exact ones (`agreement`).  `--conv-mode mfma` makes that kernel the timed one.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def _baseline_metric():
    """the metric string exactly as BASELINE.json names it"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "detections/sec (whole node), person model @640\u00d7480, 1/2/4/8 MI355X"


METRIC = _baseline_metric()
sys.path.insert(0, ROOT)

# algorithmic work of the convolution kernel per 640x480 frame (SURVEY.md section 8d / DESIGN.md):
# read features 128*C + filters, write responses 4*F*C; 2*800*F*C flop
PEAK_F32_TFLOPS = 157.3     # MI355X dense fp32 (vector = f32 MFMA) peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--conv-mode", choices=["exact", "fma", "mfma", "mfma_f16"], default="exact")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames of the same workload timed on the host CPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-other-mode", "--no-exact-ref", dest="no_other", action="store_true",
                    help="skip the run of the other convolution mode / the agreement check")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--streams", type=int, default=1,
                    help="handles (HIP streams + workspaces) the batches are fed to round-robin: batches on different handles overlap "
                         "at kernel granularity.  Pays for batches that do not fill the chip (one 640x480 frame: +30 %%, one 1080p frame: +21 %%); at the "
                         "default 64 x 640x480 it is within 1 %% and the per-kernel durations `roofline` is priced on are then those of "
                         "overlapped kernels, so the default stays 1")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 rehearsal of the multi-GPU step: initialise torch.distributed (world size 1) and issue the RCCL all_gather of the device-resident candidate payload every step")
    return ap.parse_args()


def _spawn_workers(args):
    """`python bench.py --gpus N` without a launcher: start the N workers ourselves (one process per GPU) as a CHILD
    `python -m torch.distributed.run ...` -- before this process has imported torch or touched the GPU -- forward its
    output (rank 0's JSON line goes to the inherited stdout) and exit with its return code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_workers(args))
    import torch
    import torch.distributed as dist
    from partsbaseddetector_amd import _lib, synth
    from partsbaseddetector_amd.detector import PartsBasedDetector
    from partsbaseddetector_amd.model import synthetic_person_model

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1])); s_.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    model = synthetic_person_model()
    if os.environ.get("PBD_BENCH_THRESH"):          # timing probes whose scores are deliberately wrong (tools/ab.sh): no candidates
        model.thresh = float(os.environ["PBD_BENCH_THRESH"])
    flat = model.flatten()
    B, rows, cols, cn = args.batch, args.rows, args.cols, 3
    cap = max(1 << 16, int(B * rows * cols / (480 * 640) * 1024))      # candidate capacity: ~100 per VGA frame at the synthetic threshold, 10x head-room
    # K handles = K HIP streams + K workspaces fed round-robin (detector.DetectorPool): the kernels of one batch leave parts of
    # the chip idle (tails of the distance-transform launches, launches smaller than the chip), and batches on different
    # streams fill them.  Every batch is still computed by exactly one handle; `det` (lane 0) carries the per-kernel events.
    NS = max(1, args.streams)
    conv_mode = {"exact": _lib.CONV_EXACT, "fma": _lib.CONV_FMA, "mfma": _lib.CONV_MFMA, "mfma_f16": _lib.CONV_MFMA_F16}[args.conv_mode]
    dets = []
    for _ in range(NS):
        d_ = PartsBasedDetector(device=local_rank, conv_mode=conv_mode, max_batch=B, max_candidates=cap)
        d_.distributeModel(model)
        dets.append(d_)
    det = dets[0]
    stride = det.hd.stride

    # synthetic frames, seed = global frame index + 1; resident in HBM before the timed region
    frames = np.stack([synth.synthetic_frame(rank * B + i + 1, rows, cols, cn) for i in range(B)])
    d_frames = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()

    # gather payload: [count | cap_g records] per rank; latency-bound (KBs-MBs), ONE collective per step
    from partsbaseddetector_amd import dist as pdist
    # initial gather capacity in records (7.3 MB per rank for the person model); a batch with more candidates than
    # this makes every rank send a longer prefix of its payload once -- it never raises (dist.CandidateGatherer)
    cap_g = 16384
    dev = torch.device("cuda", local_rank)
    on_device = use_dist and args.backend == "nccl"
    # one gatherer per lane (its own send / receive buffers and pending slot): every rank walks the lanes in the same order, so
    # the collectives are issued in the same order everywhere
    gatherers = [pdist.CandidateGatherer(stride, cap_g, dev if on_device else "cpu", force_collective=args.force_collective, cap_full=cap)
                 for _ in range(NS)] if use_dist else None
    gatherer = gatherers[0] if use_dist else None
    dgathers = [pdist.DeviceBatchGather(dets[i], gatherers[i]) for i in range(NS)] if on_device else None
    dgather = dgathers[0] if on_device else None
    gathered = [0]
    ncand_last = [0]

    def note(rec):
        if rec is not None:
            gathered[0] = len(rec)

    def run(steps, K=NS):
        """`steps` passes of the hot path over the resident batch, pipelined one batch deep: the host never waits for batch
        k before batch k+1 is enqueued, so the candidates' read-back (N = 1) or gather (N > 1) of batch k runs under the
        kernels of batch k+1.  Returns when every batch's candidate list is in host memory (rank 0: the gathered list)."""
        if dgather is not None:
            # N > 1 (RCCL): the candidate list never touches the host before the collective -- the walk kernel writes the
            # [found | records] payload (global frame ids), all_gather_into_tensor reads a prefix of that tensor.  Lane s % K:
            # its submit() finishes the gather of the batch that lane took K steps ago and issues this batch's collective.
            for s_ in range(steps):
                note(dgathers[s_ % K].submit(d_frames.data_ptr(), B, rows, cols, cn, frame_offset=rank * B, root_only=True))
            for i in range(min(K, steps)):                   # drain in submission order
                note(dgathers[(steps - min(K, steps) + i) % K].collect(root_only=True))
        elif gatherer is not None:
            # gloo rehearsal: host records through the pinned staging buffer
            for _ in range(steps):
                buf, n = det.detect_batch_device(d_frames.data_ptr(), B, rows, cols, cn, raw=True)
                if gatherer.pending:
                    note(gatherer.finish(root_only=True))
                gatherer.begin(buf, n, frame_offset=rank * B)
                ncand_last[0] = n
            note(gatherer.finish(root_only=True))
        else:
            # N = 1: batch s goes to lane s % K; a lane's previous batch is collected just before the lane is reused, so up to K
            # batches are in flight (K = 1: one batch enqueued ahead of the one being collected, as in rounds 1-3)
            if K == 1:
                det.submit_batch_device(d_frames.data_ptr(), B, rows, cols, cn)
                for k in range(steps):
                    if k + 1 < steps:
                        det.submit_batch_device(d_frames.data_ptr(), B, rows, cols, cn)
                    _, ncand_last[0] = det.wait_batch(raw=True)
            else:
                for s_ in range(steps):
                    if s_ >= K:
                        _, ncand_last[0] = dets[s_ % K].wait_batch(raw=True)
                    dets[s_ % K].submit_batch_device(d_frames.data_ptr(), B, rows, cols, cn)
                for i in range(min(K, steps)):
                    _, ncand_last[0] = dets[(steps - min(K, steps) + i) % K].wait_batch(raw=True)

    def sync():
        for d_ in dets:
            d_.hd.check(d_.hd.lib.pbd_synchronize(d_.hd.h))
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        run(max(args.warmup, NS))      # every lane once: plans, workspaces and pinned buffers are created on a handle's first batch
    # Inside the timed region only the dominant kernel (the convolution: one launch per step) carries HIP events -- that is
    # where `roofline` comes from.  Timing EVERY launch separates the ~45 dispatches of a step by about 9 us each
    # (profiles/r03_trace_gaps.txt: 393 us of idle GPU time per step with per-kernel events, 18 us without), so the full
    # per-kernel table (`kernel_ms_per_step`, `roofline_all`) is taken from `profiled_steps` further steps run right after
    # the timed region, and is not part of `value`.
    if not args.no_profile:
        det.hd.profile(2)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    prof_timed = det.hd.profile_read() if not args.no_profile else {}
    profiled_steps = 0
    if not args.no_profile:
        profiled_steps = max(min(args.steps, 5), 1)
        det.hd.profile(1)
        sync()
        t1 = time.perf_counter()
        run(profiled_steps, 1)          # lane 0 alone: every launch timed, kernels not overlapped with another batch's
        sync()
        dt_prof = time.perf_counter() - t1
    if dgather is not None:
        ncand_last[0] = gathered[0]
    ncand = ncand_last[0]
    rank_ms = [dt / args.steps * 1e3]
    if use_dist:
        mine = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        dt = max(float(t.item()) for t in every)          # MAX over ranks
    prof = det.hd.profile_read() if not args.no_profile else {}
    if not args.no_profile:
        det.hd.profile(0)

    if rank == 0:
        plan = det.hd.plan(rows, cols)
        cells = int(np.sum(plan["feat_rows"].astype(np.int64) * plan["feat_cols"]))
        F = flat.nfilters
        ktaps = int(flat.filter_ksize[0]) ** 2 * flat.flen
        frames_total = B * world * args.steps
        value = frames_total / dt
        # ---- algorithmic work per STEP (one batch of B frames on this rank), SURVEY.md section 8(d) / DESIGN.md section 4
        def nmix(gp):
            return int(flat.mix_offset[gp + 1] - flat.mix_offset[gp])
        # position planes are uint8 when no feature-map side exceeds 256 cells (the library picks this per plan), else int16
        pb = 1 if max(int(plan["feat_rows"].max()), int(plan["feat_cols"].max())) <= 256 else 2
        jobs, comb, parents = 0, 0, set()
        for c in range(flat.ncomponents):
            p0 = int(flat.part_offset[c])
            for gp in range(p0 + 1, int(flat.part_offset[c + 1])):
                gpar = p0 + int(flat.parentid[gp])
                K, L = nmix(gp), nmix(gpar)
                jobs += K                                   # (part, mixture) distance transforms per level
                comb += K * 4 + L * 1                       # per child: dt in, Ik out (Ix / Iy are composed lazily from the transform's planes)
                parents.add(gpar)
        comb += sum(8 * nmix(g) for g in parents)           # per parent: response in, accumulated score out
        work = {
            "k_conv": {"bytes": (128 * cells + 4 * F * ktaps + 4 * F * cells) * B, "flop": 2.0 * ktaps * F * cells * B},
            "k_dt_rows": {"bytes": (8 + pb) * cells * jobs * B},      # read score 4, write tmp 4 + Ix
            "k_dt_cols": {"bytes": (8 + pb) * cells * jobs * B},      # read tmp 4, write dt 4 + Iy (the rows pass writes Ix in place)
            "k_dp_combine": {"bytes": comb * cells * B},        # per child: dt in, Ik out; per parent: score in/out
            "k_hog_hist": {"bytes": (3 * int(np.sum(plan["img_rows"].astype(np.int64) * plan["img_cols"])) + 76 * cells) * B},
        }
        psteps = max(profiled_steps, 1)
        stage_ms = {k: round(ms / psteps, 4) for k, (ms, n) in prof.items() if n}
        roofline, roof_all = None, []
        traffic_tab = {}
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        # the counters were collected on the default workload (64 frames of 640x480): other shapes carry no traffic figure
        if os.path.exists(tf) and (B, rows, cols) == (64, 480, 640):
            try:
                traffic_tab = json.load(open(tf))
            except Exception:
                traffic_tab = {}
        for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
            if not n or k not in work:
                continue
            per_step = n / psteps                           # launches per step (depth groups, chunks)
            if k == "k_conv" and prof_timed.get(k, (0, 0))[1]:
                ms, n = prof_timed[k]                       # the dominant kernel: measured live over the TIMED region
            avg_s = ms / n * 1e-3
            by = work[k]["bytes"] / per_step
            tr = traffic_tab.get(k, {}).get("hbm_bytes_per_step")
            symbol = {"k_conv": "k_conv_mfma" if args.conv_mode in ("mfma", "mfma_f16") else "k_conv3", "k_hog_hist": "k_hog_tile"}.get(k, k)
            entry = {"kernel": k, "symbol": symbol, "avg_launch_ms": round(ms / n, 4), "launches_per_step": per_step,
                     "algorithmic_bytes_per_launch": int(by), "traffic": (tr / per_step) if tr else None}
            if k == "k_conv":
                fl = work[k]["flop"] / per_step
                mfma = args.conv_mode in ("mfma", "mfma_f16")
                peak = PEAK_BF16_TFLOPS if mfma else PEAK_F32_TFLOPS
                ach = fl / avg_s / 1e12
                entry.update({"bound": "mfma" if mfma else "valu_fp32", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                              "algorithmic_flop_per_launch": fl, "hbm_achieved_GBps": round(by / avg_s / 1e9, 2),
                              "note": ("fp32-equivalent flops (2*800*F*cells); the kernel executes 3 bf16 MFMAs per product tile, "
                                       "i.e. 3x these flops on the matrix cores, priced against the dense bf16 peak") if mfma else
                                      ("fp32 contraction, 330 flop/B: compute-bound; exact mode must round multiply and add "
                                       "separately (2 VALU lane-ops per MAC), priced against the dense fp32 FMA peak")})
            else:
                ach = by / avg_s / 1e9
                entry.update({"bound": "hbm", "achieved": round(ach, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                              "frac": round(ach / PEAK_HBM_GBPS, 4)})
                if k in ("k_dt_rows", "k_dt_cols"):
                    entry["note"] = ("priced against HBM as SURVEY 8(d) prescribes, but the passes are bound by VALU issue (70-82 % busy): 200 instructions per element "
                                     "and wave where the divergence-free skeleton needs 90 -- 64 independent rows in lock step run 2.45 pop iterations per element "
                                     "where a lane needs 0.71.  Plane I/O plus the MAXIMUM possible spill traffic costs 3.5-4 ms per pass (uniform-control-flow "
                                     "probes and counters: profiles/r03_dt/README.md section 6); launches that do not fill the chip run as narrower waves (section 7)")
            entry["wasted_traffic"] = round(entry["traffic"] / entry["algorithmic_bytes_per_launch"], 3) if entry["traffic"] else None
            roof_all.append(entry)
        if roof_all:
            roofline = roof_all[0]                           # the dominant kernel of this run
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (bench contract)
            # the GPU box's CPU share for one GPU is 16 cores; use at most that many OpenMP threads
            ncpu = min(len(os.sched_getaffinity(0)), 16)
            from oracle import oracle   # CPU restatement: the reported baseline, never the measured path
            oracle.build()
            oracle.set_num_threads(ncpu)
            nf = max(1, args.cpu_frames)
            oracle.detect(flat, frames[0])            # warm-up (page-in, thread pool)
            t1 = time.perf_counter()
            stage = {}
            for i in range(nf):
                _, ms = oracle.detect(flat, frames[i % B], want_stage_ms=True)
                for k, v in ms.items():
                    stage[k] = stage.get(k, 0.0) + v / nf
            cdt = time.perf_counter() - t1
            # one frame on a single thread as well ("time 1 thread and all cores")
            oracle.set_num_threads(1)
            t1 = time.perf_counter()
            oracle.detect(flat, frames[0])
            c1 = time.perf_counter() - t1
            oracle.set_num_threads(ncpu)
            # the same sources at -O3, in a child interpreter (oracle.py's PBD_ORACLE_SO switch): SURVEY 8(d) "also -O3"
            o3 = None
            try:
                import subprocess
                so3 = os.path.join(ROOT, "oracle", "libpbd_oracle_o3.so")
                subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), so3])
                code = ("import sys, time, numpy as np; sys.path.insert(0, %r)\n"
                        "from oracle import oracle; from partsbaseddetector_amd import synth\n"
                        "from partsbaseddetector_amd.model import synthetic_person_model\n"
                        "flat = synthetic_person_model().flatten(); oracle.set_num_threads(%d)\n"
                        "fr = [synth.synthetic_frame(i + 1, %d, %d, 3) for i in range(%d)]\n"
                        "oracle.detect(flat, fr[0]); t = time.perf_counter()\n"
                        "for f in fr: oracle.detect(flat, f)\n"
                        "print(len(fr) / (time.perf_counter() - t))\n") % (ROOT, ncpu, rows, cols, min(nf, 4))
                r3 = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                                    env=dict(os.environ, PBD_ORACLE_SO=so3))
                o3 = {"value": round(float(r3.stdout.strip().splitlines()[-1]), 4), "unit": "detections/s", "cores": ncpu,
                      "flags": "-O3 -ftree-vectorize -msse4.1 -fopenmp -ffp-contract=off", "sample": f"{min(nf, 4)} frames"}
            except Exception as e:          # the baseline is context, never a reason to lose the bench line
                o3 = {"error": str(e)[:200]}
            cpu = {"value": round(nf / cdt, 4), "unit": "detections/s", "cores": oracle.num_threads(), "kind": "port",
                   "flags": "-O2 -ftree-vectorize -msse4.1 -fopenmp -ffp-contract=off (the reference's RelWithDebInfo, CMakeLists.txt:56-61,74-81)",
                   "O3": o3,
                   "sample": f"{nf} of the same {cols}x{rows} frames, full path, OpenMP at the reference's 5 sites",
                   "stage_ms_per_frame": {k: round(v, 2) for k, v in stage.items()},
                   "single_thread": {"value": round(1.0 / c1, 4), "unit": "detections/s", "cores": 1, "sample": "1 frame"}}
        other_mode, agreement = None, None
        if world == 1 and not use_dist and args.conv_mode in ("exact", "mfma") and not args.no_other:
            # the other convolution mode on the same resident batch: exact <-> matrix cores
            oname = "mfma" if args.conv_mode == "exact" else "exact"
            last = np.array(det._buf[:ncand * stride]).reshape(ncand, stride).copy()
            det2 = PartsBasedDetector(device=local_rank, conv_mode=_lib.CONV_MFMA if oname == "mfma" else _lib.CONV_EXACT,
                                      max_batch=B, max_candidates=cap)
            det2.distributeModel(model)
            buf2, n2 = det2.detect_batch_device(d_frames.data_ptr(), B, rows, cols, cn, raw=True)      # warm-up
            det2.hd.check(det2.hd.lib.pbd_synchronize(det2.hd.h))
            t1 = time.perf_counter()
            for _ in range(3):
                buf2, n2 = det2.detect_batch_device(d_frames.data_ptr(), B, rows, cols, cn, raw=True)
            det2.hd.check(det2.hd.lib.pbd_synchronize(det2.hd.h))
            edt = (time.perf_counter() - t1) / 3
            ref = np.array(buf2[:n2 * stride]).reshape(n2, stride)
            # record-by-record comparison keyed by (frame, component, level, root x, root y)
            ka = {tuple(r[:5]): r for r in last}
            kb = {tuple(r[:5]): r for r in ref}
            common = set(ka) & set(kb)
            boxes_same = sum(int(np.array_equal(ka[k][6:], kb[k][6:])) for k in common)
            sdiff = max((abs(float(ka[k][5:6].view(np.float32)[0]) - float(kb[k][5:6].view(np.float32)[0])) for k in common), default=0.0)
            odd = [r for k, r in list(ka.items()) + list(kb.items()) if k not in common]
            # a root present in only one mode must sit within the score tolerance of the threshold
            margin = max((abs(float(r[5:6].view(np.float32)[0]) - flat.thresh) for r in odd), default=0.0)
            other_mode = {"conv_mode": oname, "value": round(B / edt, 3), "unit": "detections/s", "ms_per_step": round(edt * 1e3, 3), "streams": 1,
                          "note": ("matrix-core convolution (bf16 hi/lo operand split, fp32 accumulation): responses within 1e-4, "
                                   "index outputs can differ on near ties (see agreement)") if oname == "mfma" else
                                  "bit-identical responses (reference summation order)"}
            agreement = {"candidates_exact": int(ncand if oname == "mfma" else n2), "candidates_mfma": int(n2 if oname == "mfma" else ncand),
                         "common": len(common), "common_with_identical_parts": int(boxes_same), "only_in_one_mode": len(odd),
                         "max_score_diff_common": sdiff, "max_threshold_margin_of_unmatched": margin, "tolerance": 1e-4,
                         "records_identical": bool(len(odd) == 0 and boxes_same == len(common))}
            det2.hd.close()
        host_input = None
        if world == 1 and not use_dist and not args.no_other:
            # SURVEY 8(d)'s end-to-end form of the metric: frames in (pageable) HOST memory -> candidate lists in host
            # memory.  The pipelined entry points keep two batches in flight, so the host-side staging and the PCIe copy of
            # batch k+1 overlap the kernels of batch k.  Reported beside `value` (which, per the bench contract, is
            # measured with the frames resident in HBM), never as `value`.
            import ctypes as C
            fr = [np.ascontiguousarray(frames[i]) for i in range(B)]
            hb, hn = np.zeros(cap * stride, np.int32), C.c_int()
            lib, hh = det.hd.lib, det.hd.h
            def submit():
                det.hd.check(lib.pbd_detect_batch_submit(hh, B, _lib.ptr_array(fr), rows, cols, cn, cols * cn))
            def wait():
                det.hd.check(lib.pbd_detect_batch_wait(hh, hb.ctypes.data, cap, C.byref(hn)))
            submit(); wait()                                   # warm-up: staging buffers
            K = max(args.steps, 4)
            t1 = time.perf_counter()
            submit()
            for k in range(K):
                if k + 1 < K:
                    submit()
                wait()
            pdt = (time.perf_counter() - t1) / K
            same = int(hn.value) == int(ncand) and np.array_equal(hb[:ncand * stride], np.asarray(det._buf[:ncand * stride]))
            # and the synchronous host entry point (one batch at a time, copy not overlapped)
            def hstep():
                det.hd.check(lib.pbd_detect_batch(hh, B, _lib.ptr_array(fr), rows, cols, cn, cols * cn, hb.ctypes.data, cap, C.byref(hn)))
            hstep()
            t1 = time.perf_counter()
            for _ in range(3):
                hstep()
            hdt = (time.perf_counter() - t1) / 3
            host_input = {"value": round(B / pdt, 3), "unit": "detections/s", "ms_per_step": round(pdt * 1e3, 3),
                          "vs_device_resident": round((B / pdt) / value, 4), "records_identical_to_device_run": bool(same),
                          "note": "host -> host: pbd_detect_batch_submit / _wait, two batches in flight (pinned staging + H2D of "
                                  "batch k+1 on a copy stream under the kernels of batch k)",
                          "synchronous": {"value": round(B / hdt, 3), "ms_per_step": round(hdt * 1e3, 3),
                                          "note": "pbd_detect_batch: pageable host pointers, copy inside the call, not overlapped"}}
        out = {
            "metric": METRIC, "value": round(value, 3),
            "unit": "detections/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{3 if (rows, cols) == (1080, 1920) else 2}]: synthetic person model (26 parts x 6 mixtures = 156 filters "
                                   f"5x5x32), batch of {B} {cols}x{rows} frames per GPU, full HOG+conv+DT/DP+argmin on GPU",
                       "frames_per_gpu_per_step": B, "conv_mode": args.conv_mode, "candidates_last_step": int(gathered[0] if world > 1 else ncand),
                       "parallelism": f"frames sharded over {world} GPU(s), RCCL all_gather of candidates",
                       "world_size": dist.get_world_size() if use_dist else 1,
                       "backend": dist.get_backend() if use_dist else None,
                       "streams": NS,
                       "pipeline": (f"{NS} handles (HIP streams + workspaces) fed round-robin, up to {NS} batches in flight: a handle's candidate list is "
                                    "collected just before the handle is given its next batch" if NS > 1 else
                                    "one batch deep: batch k+1 is enqueued before the candidate list of batch k is collected"),
                       "rank_ms_per_step": [round(v, 3) for v in rank_ms],
                       "gather": ({"collectives_per_step": sum(g_.collectives for g_ in gatherers) / max(args.steps + (max(args.warmup, NS) if args.warmup > 0 else 0) + profiled_steps, 1),
                                   "capacity_records": gatherer.cap, "grown": gatherer.grown,
                                   "payload": "device-resident: written by the walk kernel, handed to all_gather_into_tensor as it is" if on_device else "host records (gloo rehearsal)",
                                   "overlap": "the collective of batch k runs under the kernels of batch k+1 (begin / finish one step apart)"} if gatherer else None)},
            "roofline": roofline, "cpu_baseline": cpu, "kernel_ms_per_step": stage_ms,
            "kernel_timing": ({"roofline": "k_conv: HIP events on its launch inside the timed region (avg of %d launches)" % (prof_timed.get("k_conv", (0, 0))[1]),
                               "kernel_ms_per_step": "HIP events on every launch, %d further steps right after the timed region (%.3f ms per step there)" % (profiled_steps, dt_prof / psteps * 1e3),
                               "step_minus_kernels_ms": round(dt / args.steps * 1e3 - sum(stage_ms.values()), 3),
                               "note": "step_minus_kernels_ms compares back-to-back execution with kernels timed in isolation (each dispatch then starts on an idle chip "
                                       "and runs 1-2 % faster), so it overstates the idle time; the rocprofv3 kernel trace of the timed configuration shows 18 us of idle "
                                       "GPU time per 64-frame step (profiles/r03_trace_gaps.txt)"} if not args.no_profile else None),
            "roofline_all": roof_all,
            ("fast_mode" if args.conv_mode == "exact" else "exact_mode"): other_mode, "agreement": agreement,
            "host_input": host_input,
        }
        print(json.dumps(out), flush=True)
    for d_ in dets:
        d_.hd.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
