"""In-tree build of the HIP library (libpbd_hip.so) for gfx950.

hipcc cross-compiles without a GPU.  Flags that matter for parity:
  -ffp-contract=off                           no fused multiply-add unless written as one
  -fhip-fp32-correctly-rounded-divide-sqrt    IEEE fp32 divide / sqrt
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpbd_hip.so")
SOURCES = ["pbd_capi.hip", "pbd_kernels_features.hip", "pbd_kernels_conv.hip", "pbd_kernels_conv_mfma.hip", "pbd_kernels_dp.hip"]
HEADERS = ["pbd_internal.h", os.path.join("..", "..", "include", "pbd.h")]
FLAGS = os.environ.get("PBD_EXTRA_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall"]
OBJ = os.path.join(CSRC, "build")          # object files (git-ignored); one per source so that they compile in parallel


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP library cannot be built")
    return exe


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(obj)
        spath = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), hdr_t):
            continue
        cmd = [hipcc()] + FLAGS + ["-c", "-o", obj, spath]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
