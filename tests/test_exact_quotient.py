"""The division-free quotient of the StringLength fast kernel (csrc/stringlength.hip,
fast::exact_quotient: correctly rounded reciprocal + two fma corrections) equals the IEEE division
t / period - checked here on the CPU with the same operation sequence (x86 fma), on random and
adversarial operands.  The GPU parity tests compare the kernel's sort order with numpy's division
end to end; this test isolates the arithmetic identity."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_two_fma_corrections_reproduce_the_ieee_quotient(tmp_path):
    flags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    if " fma" not in flags:
        pytest.skip("host CPU has no fma instruction")
    exe = tmp_path / "exact_quotient_check"
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(HERE, "csrc", "exact_quotient_check.c"),
                    "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe), "20000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "two_step_mismatches 0" in out.stdout, out.stdout
