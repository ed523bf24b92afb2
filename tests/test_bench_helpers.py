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


def test_shared_axis_entry_finds_whichever_shared_kernel_ran(tmp_path, monkeypatch):
    """Round-5 defect: the c3_shared_t entry looked up "gls_shared_kernel" while `gls_shared2_kernel` is what runs
    (individual weights), so its issue fraction came back None.  The lookup key is now the common prefix."""
    mix = {c: 10 for c in bench.F64_COUNTERS}
    summ = {"src_sha": bench.source_hashes(), "kernels": {
        "gls_shared2_kernel<true> grid=12812288": dict(mix, SQ_INSTS_VALU=40_000_000_000, ms=88.0),
        "gls_scan_kernel<16, 0, 2, true> grid=131072": dict(mix, SQ_INSTS_VALU=13_000_000_000, ms=27.0)}}
    path = tmp_path / "pmc.json"
    path.write_text(json.dumps(summ))
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(path))
    fr, blk = bench.two_fracs("gls_shared", 87.0, 2.0, "unit")
    assert blk is not None and blk["kernel"].startswith("gls_shared2_kernel") and fr["executed_issue_frac"] is not None
    assert bench.two_fracs("gls_shared_kernel", 87.0, 2.0, "unit")[1] is None      # the old key: no match


def test_executed_flops_are_counted_from_the_typed_counters():
    blk = {"mix": {"fma_f64": 12_651_000_000, "add_f64": 486_000_000, "mul_f64": 9_000_000}}
    got = bench.executed_flops(blk, 26.35)
    assert abs(got["TFLOPs"] - (2 * 12.651e9 + 0.486e9 + 0.009e9) * 64 / 26.35e-3 / 1e12) < 0.01     # 62.7 (round-5 verdict)
    assert 0.79 < got["frac_of_peak"] < 0.81
    assert bench.executed_flops(None, 1.0) is None and bench.executed_flops({"mix": {"fma_f64": 1}}, 1.0) is None


def test_fft_path_roofline_counts_passes_and_reads_the_traffic(tmp_path, monkeypatch):
    r = bench.fft_path_roofline(100_000, 1_000_000, 0.5)
    assert r["bound"] == "hbm" and r["nfft"] == 1 << 23 and r["passes_per_grid"] == 3 and r["grids"] == 3
    assert 1.5e9 < r["algorithmic_bytes"] < 2.2e9 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    summ = {"src_sha": bench.source_hashes(), "kernels": {
        "fft_pass_lds_kernel<16> grid=524288": {"hbm_bytes": 200_000_000, "dispatches_per_pass": 36, "ms": 0.046},
        "glsfft_epilogue_kernel grid=1000192": {"hbm_bytes": 50_000_000, "dispatches_per_pass": 6, "ms": 0.02},
        "glsfft_epilogue_kernel grid=4096": {"hbm_bytes": 1_000, "dispatches_per_pass": 5, "ms": 0.002},
        "gls_scan_kernel<16, 0, 2, true> grid=131072": {"hbm_bytes": 1, "dispatches_per_pass": 6, "ms": 27.0}}}
    path = tmp_path / "pmc.json"
    path.write_text(json.dumps(summ))
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(path))
    r = bench.fft_path_roofline(100_000, 1_000_000, 0.5)
    assert r["traffic"] == 6 * 200_000_000 + 50_000_000          # per launch: 36 / 6 pass dispatches + one epilogue
    summ["src_sha"]["glsfft.hip"] = "0" * 16
    path.write_text(json.dumps(summ))
    r = bench.fft_path_roofline(100_000, 1_000_000, 0.5)
    assert r["traffic"] is None and "stale" in r["traffic_note"]
