"""bench.py's roofline block reads executed-instruction counts from profiles/r06_pmc_summary.json and
refuses entries collected for other kernel sources.  This test keeps the committed summary in step
with the committed sources of the HEADLINE kernel (re-run tools/collect_pmc.sh on a GPU box after
touching them), and checks the refusal logic itself."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_pmc_summary_matches_the_headline_kernel_sources():
    summ = json.load(open(bench.PMC_SUMMARY))
    now = bench.source_hashes()
    for name in bench.sources_of("gls_scan_kernel<16, 0, 2>"):
        assert summ["src_sha"].get(name) == now[name], f"{name} changed since the PMC passes: re-collect"
    k, why = bench.pmc_for("gls_scan_kernel", 27.5)
    assert why is None and k["SQ_INSTS_VALU"] > 1e10
    blk, _ = bench.valu_issue_block("gls_scan_kernel", 27.5)
    assert 0.5 < blk["frac"] <= blk["frac_all_at_4_cycles"] <= 1.0   # executed-issue fractions, never above 1
    assert blk["mix"]["fp64_share"] > 0.8                # the headline kernel is nearly pure fp64 arithmetic


def test_stale_or_missing_entries_are_refused(tmp_path, monkeypatch):
    summ = json.load(open(bench.PMC_SUMMARY))
    summ["src_sha"]["gls.hip"] = "0" * 16
    path = tmp_path / "stale.json"
    path.write_text(json.dumps(summ))
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(path))
    k, why = bench.pmc_for("gls_scan_kernel", 27.5)
    assert k is None and "gls.hip" in why and "stale" in why
    k, why = bench.pmc_for("gls_scan_kernel", 500.0)     # no profiled launch of that duration
    assert k is None and "within 25%" in why
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(tmp_path / "absent.json"))
    k, why = bench.pmc_for("gls_scan_kernel", 27.5)
    assert k is None and why.endswith("absent.json is missing")


def test_l2_gather_ceiling_is_read_from_a_hashed_profile(tmp_path, monkeypatch):
    """The StringLength roofline's ceiling comes from profiles/r03_ubench_gather_rate.json (written by
    tools/ubench_summary.py with the sha256 of tools/ubench/gather_rate.hip), not from a constant."""
    ceiling, why = bench.l2_gather_ceiling()
    assert why is None and 1e11 < ceiling < 1e12
    stale = json.load(open(bench.GATHER_UBENCH))
    stale["src_sha"] = "0" * 16
    path = tmp_path / "stale.json"
    path.write_text(json.dumps(stale))
    monkeypatch.setattr(bench, "GATHER_UBENCH", str(path))
    ceiling, why = bench.l2_gather_ceiling()
    assert ceiling is None and "stale" in why
