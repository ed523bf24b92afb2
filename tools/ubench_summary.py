"""Summarise a run of tools/ubench/gather_rate.hip (its text output) as profiles/rNN_ubench_gather_rate.json:
the L2 random-gather ceiling bench.py prices the StringLength kernel against, with the sha256 of the
micro-benchmark's source so that bench.py refuses a figure measured with another version of it.

    python tools/ubench_summary.py gpurun_out/r03x/gather_rate.txt profiles/r03_ubench_gather_rate.json
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    text = open(sys.argv[1]).read()
    rates = {"4": [], "8": [], "16": []}
    for line in text.splitlines():
        m = re.match(r"random gather (\d+) B from .*?([\d.]+) Gaccess/s chip", line)
        if m:
            rates[m.group(1)].append(float(m.group(2)) * 1e9)
    assert rates["16"], "no 'random gather 16 B' lines found"
    src = os.path.join(ROOT, "tools", "ubench", "gather_rate.hip")
    out = {"src_sha": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
           "gather_16B_per_s": float(np.median(rates["16"])),
           "gather_8B_per_s": float(np.median(rates["8"])) if rates["8"] else None,
           "gather_4B_per_s": float(np.median(rates["4"])) if rates["4"] else None,
           "samples_16B": rates["16"],
           "note": "median over the 'random gather 16 B from 800 KB (AoS t,m)' lines of tools/ubench/gather_rate.hip "
                   "(1024-thread workgroups, one per CU, indices from an LDS-resident random permutation): accesses "
                   "per second chip-wide into an L2-resident table",
           "raw": os.path.basename(sys.argv[1])}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(f"wrote {sys.argv[2]}: {out['gather_16B_per_s'] / 1e9:.1f} G gathers/s")


if __name__ == "__main__":
    main()
