"""CPU sanitizer leg: the oracle's C restatement built with
AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile, target `sanitize`) runs the whole
golden-vector suite in a child interpreter with libasan preloaded; any out-of-bounds access,
use-after-free, signed overflow, misaligned access or invalid shift in the restatement aborts
the child.  (The oracle is what every GPU parity test leans on, so it is the C code worth
sanitizing; sanitizers never run on the GPU.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_vectors_under_asan_ubsan():
    if shutil.which("gcc") is None or shutil.which("make") is None:
        pytest.skip("no gcc / make")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan is not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "_san", "librmh_oracle_san.so")
    env = dict(os.environ, LD_PRELOAD=asan, RMH_ORACLE_SO=so, OMP_NUM_THREADS="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        "-q", "-x", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
