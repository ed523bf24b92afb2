"""The C restatement of the oracle (oracle/scan_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer:
`make -C oracle SAN=1`, then the golden suite's C legs in a child interpreter with libasan preloaded.
CPU only (the GPU pool has no sanitizer runs); test infrastructure checking test infrastructure."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_is_clean_under_asan_and_ubsan():
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "SAN=1"], check=True)
    env = dict(os.environ,
               LD_PRELOAD=os.path.realpath(asan),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               ORACLE_LIBRARY=os.path.join(ROOT, "oracle", "_build", "liboracle_san.so"),
               OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = out.stdout[-1500:] + out.stderr[-1500:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
