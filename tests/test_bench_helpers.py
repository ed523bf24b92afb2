"""CPU checks of bench.py's host-side helpers: the slab arithmetic of the sharded extras, the kernel matching
of the two roofline fields, and the CPU-baseline child process (multiprocessing.Pool as upstream,
/root/reference/src/periodicity/phase.py:69-70,185-186) at a toy size."""
import json
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_covers_the_grid_once():
    for total in (0, 1, 7, 4096, 100_000):
        for n in (1, 2, 3, 8):
            cuts = [bench.slab(total, n, i) for i in range(n)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert max(e - b for b, e in cuts) <= -(-total // n) if total else True


def test_two_fracs_match_kernels_by_every_given_substring(tmp_path, monkeypatch):
    summ = {"src_sha": bench.source_hashes(), "kernels": {
        "pdm_scan_kernel<256, 4, true, 0> grid=1200384": {"SQ_INSTS_VALU": 1_000_000_000, "ms": 2.5, "GRBM_GUI_ACTIVE": 44_000_000},
        "pdm_scan_kernel<256, 4, true, 1> grid=1200384": {"SQ_INSTS_VALU": 1_100_000_000, "ms": 2.6},
        "pdm_scan_kernel<256, 4, false, 2> grid=400128": {"SQ_INSTS_VALU": 1_400_000_000, "ms": 2.8}}}
    path = tmp_path / "pmc.json"
    path.write_text(json.dumps(summ))
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(path))
    fr, blk = bench.two_fracs(("pdm_scan_kernel<", ", 1> grid"), 2.6, 0.9, "unit")
    assert blk["kernel"].endswith(", 1> grid=1200384") and fr["algorithmic_frac"] == 0.9
    assert abs(fr["executed_issue_frac"] - 1.1e9 * 4 / 1024 / 2.4e9 / 2.6e-3) < 1e-3
    fr, blk = bench.two_fracs(("pdm_scan_kernel<", ", 0> grid"), 2.5, 1.0, "unit")
    assert "profiled_clock_GHz" in fr["executed_issue"] and 2.0 < fr["executed_issue"]["profiled_clock_GHz"] < 2.5
    fr, blk = bench.two_fracs("no_such_kernel", 1.0, None, "unit")
    assert blk is None and fr["executed_issue_frac"] is None and "no profiled" in fr["executed_issue_note"]


def test_cpu_pool_baseline_child_process_runs():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_pool_baseline.py"), "16", "2"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-1500:]
    res = json.loads(run.stdout.strip().splitlines()[-1])
    assert res["cores"] == 2 and res["periods_sampled"] == 16
    for key in ("pdm", "stringlength"):
        assert res[key]["map_s"] > 0 and res[key]["Gpair_per_s"] > 0
