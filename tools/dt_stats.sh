#!/bin/bash
# Lock-step statistics of the distance-transform passes.  Build the instrumented library HERE (cross-compile):
#     tools/dt_stats.sh build
# then on the GPU box:  tools/dt_stats.sh run   -> gpurun_out/dt_stats.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=$R/partsbaseddetector_amd/csrc/build
if [ "$1" = build ]; then
    FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
    /opt/rocm/bin/hipcc $FL -DPBD_DT_STATS -c -o $B/pbd_kernels_dp_stats.o $R/partsbaseddetector_amd/csrc/pbd_kernels_dp.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libpbd_dtstats.so $B/pbd_capi.o $B/pbd_kernels_features.o $B/pbd_kernels_conv.o $B/pbd_kernels_conv_mfma.o $B/pbd_kernels_dp_stats.o
    exit $?
fi
# usage on the box: tools/dt_stats.sh run [rows cols frames]   (default 480 640 4)
ROWS=${2:-480}; COLS=${3:-640}; NF=${4:-4}
cd $R && PBD_LIB=$B/libpbd_dtstats.so DT_ROWS=$ROWS DT_COLS=$COLS DT_NF=$NF timeout -k 10 600 python3 - <<'PY' | tee $R/gpurun_out/dt_stats_${ROWS}x${COLS}.txt
import ctypes, numpy as np
from partsbaseddetector_amd import _lib, detector, model as M, synth
lib = ctypes.CDLL(_lib.LIB_PATH)
import os
rows, cols, nf = int(os.environ["DT_ROWS"]), int(os.environ["DT_COLS"]), int(os.environ["DT_NF"])
det = detector.PartsBasedDetector(device=0, max_batch=nf)
det.distributeModel(M.synthetic_person_model(thresh=18.9))
frames = [synth.synthetic_frame(100 + i, rows, cols, 3) for i in range(nf)]
print(f"{nf} frames of {cols}x{rows}")
out = (ctypes.c_ulonglong * 16)()
lib.pbd_debug_dt_stats(out, 1)
det.detect_batch(frames)
lib.pbd_debug_dt_stats(out, 0)
names = ["forward: elements", "forward: pop iterations", "read-out: pop iterations", "push: spill path", "pop: reload path", "read-out: elements", "forward: undecided pre-tests (exact path)"]
for i, n in enumerate(names):
    w, l = out[2 * i], out[2 * i + 1]
    print(f"{n:28s} wave executions {w:12d}  lanes {l:14d}  lanes/execution {l / max(w, 1):6.2f}")
fw, fl = out[0], out[1]
print("per forward element (wave level): pop iterations %.3f (lane level %.3f), spill path %.3f (lane %.3f), reload path %.3f (lane %.3f); read-out pop iterations %.3f (lane %.3f)" % (
    out[2] / fw, out[3] / fl, out[6] / fw, out[7] / fl, out[8] / fw, out[9] / fl, out[4] / out[10], out[5] / out[11]))
PY
