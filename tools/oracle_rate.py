"""TEST-INFRASTRUCTURE timing: throughput of the C checkers (oracle/scan_oracle.c) on this host by thread count,
to size the exhaustive parity tests.  `python tools/oracle_rate.py [threads ...]` - each count in a child process."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, numpy as np
sys.path.insert(0, %r)
from oracle import c_oracle as co
rng = np.random.default_rng(1)
n, nf = 100000, 20000
t = np.sort(rng.uniform(0, n, n)); dy = rng.uniform(.05, .2, n); y = 1 + .5 * np.sin(2 * np.pi * t / 37.3) + dy * rng.standard_normal(n)
freq = 0.5 / n / 5 + np.arange(nf) / n / 5
co.gls_power_f64(t, y, dy, freq[:512])
t0 = time.time(); co.gls_power_f64(t, y, dy, freq); dt = time.time() - t0
print("threads", sys.argv[1], "gls f64 N=1e5 x 2e4: %%.2f Gpair/s" %% (n * nf / dt / 1e9), flush=True)
n = 1000000; nf = 4000
t = np.sort(rng.uniform(0, n, n)); dy = rng.uniform(.05, .2, n); y = 1 + .5 * np.sin(2 * np.pi * t / 37.3) + dy * rng.standard_normal(n)
freq = 0.5 / n / 5 + np.arange(nf) / n / 5
t0 = time.time(); co.gls_power_f64(t, y, dy, freq); dt = time.time() - t0
print("threads", sys.argv[1], "gls f64 N=1e6 x 4e3: %%.2f Gpair/s" %% (n * nf / dt / 1e9), flush=True)
"""

if __name__ == "__main__":
    for th in (sys.argv[1:] or [str(os.cpu_count())]):
        env = dict(os.environ, OMP_NUM_THREADS=th, OMP_PROC_BIND="false")
        subprocess.run([sys.executable, "-c", CHILD % ROOT, th], env=env, check=False)
