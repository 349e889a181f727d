"""the CPU restatement is run under AddressSanitizer + UBSan (GPU ASan is not available on this pool;
sanitizers run on the CPU build only).  oracle/Makefile's `libpbd_oracle_asan.so` target (-O1 -g -fsanitize=address,undefined)
is loaded -- through oracle.py's PBD_ORACLE_SO switch, with libasan preloaded into a child interpreter -- and the oracle's own
tests (golden vectors + independent formulations) are repeated on it: they must pass and the sanitizers must stay silent."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_and_ubsan():
    so = os.path.join(ROOT, "oracle", "libpbd_oracle_asan.so")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), so], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan is not installed")
    env = dict(os.environ, PBD_ORACLE_SO=so, LD_PRELOAD=libasan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                        os.path.join(ROOT, "tests", "test_golden.py"), os.path.join(ROOT, "tests", "test_oracle_cpu.py")],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-3000:]
    assert r.returncode == 0, out[-3000:]
    assert " passed" in r.stdout
