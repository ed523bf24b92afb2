"""The C-ABI library loads and exports every symbol include/periodicity_hip.h declares.
No compute call is made (there is no GPU where this runs)."""
import ctypes
import os
import re

import pytest

from periodicity_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "periodicity_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdc_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    if not os.path.isfile(_cabi.library_path()):
        import __graft_entry__ as entry
        entry.build()
    return ctypes.CDLL(_cabi.library_path())


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_cabi.PROTOTYPES)


def test_every_declared_symbol_is_exported(built):
    missing = [s for s in declared_symbols() if not hasattr(built, s)]
    assert not missing, missing


def test_runtime_entry_points_without_a_gpu(built):
    lib = _cabi.lib()
    assert lib.pdc_version() >= 1
    assert isinstance(lib.pdc_last_error(), bytes)
    assert lib.pdc_gls_work_bytes(1000, 1, 1000) >= 48000
    assert lib.pdc_gls_work_bytes(-1, 1, 10) == -1
    if _cabi.device_count() == 0:
        # the product path must fail loudly, never fall back to the CPU
        import numpy as np
        with pytest.raises((RuntimeError, ValueError)):
            _cabi.gls_scan(np.arange(4.0), np.ones(4), None, 0.1, 0.1, 4)


def test_integration_md_binding_snippet_runs_against_the_built_library(built, monkeypatch):
    """The ctypes stub INTEGRATION.md tells a maintainer of the reference to add must keep loading:
    every symbol it names exists, its argtypes marshal, and without a GPU the calls fail loudly with
    the library's own error (never a crash, never a CPU answer)."""
    import numpy as np
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    snippet = next(b for b in blocks if "C.CDLL" in b)
    monkeypatch.setenv("PERIODICITY_HIP_LIB", _cabi.library_path())
    ns = {}
    exec(compile(snippet, "INTEGRATION.md:_hip.py", "exec"), ns)
    for name in re.findall(r"_lib\.(pdc_[a-z0-9_]+)", snippet):
        assert name in _cabi.PROTOTYPES, name
        if name != "pdc_last_error":        # same argument list as the package's own binding
            assert len(getattr(ns["_lib"], name).argtypes) == len(_cabi.PROTOTYPES[name][1]), name
    t = np.arange(8.0)
    calls = [lambda: ns["gls_power"](t, np.ones(8), None, 0.1 + 0.1 * np.arange(4), True, False),
             lambda: ns["trig_sum"](t, np.ones(8), 0.1, 4, 0.05),
             lambda: ns["pdm_thetas"](t, np.cos(t), [2.0, 3.0], 5, 2, 0.5),
             lambda: ns["string_lengths"](t, np.cos(t) / 4, [2.0, 3.0])]
    if _cabi.device_count() == 0:
        for call in calls:
            with pytest.raises(RuntimeError):
                call()
    else:
        for call in calls:
            assert np.all(np.isfinite(call()))


def test_workspace_budget_scales_the_size_functions_without_a_gpu(built, monkeypatch):
    """PDC_WORK_BUDGET_GB (round 6) is read by the size functions themselves - no device involved -, so the rule can be
    checked here: the built-in caps ask for tens of GB at a million samples, a budget holds the answer under itself,
    a smaller budget gives a smaller workspace, and without the variable the old sizes come back."""
    lib = _cabi.lib()
    n, n_per = 1_000_000, 2048
    monkeypatch.delenv("PDC_WORK_BUDGET_GB", raising=False)
    free_sl, free_ss = lib.pdc_stringlength_work_bytes(n, n_per), lib.pdc_supersmoother_work_bytes(n, n_per)
    assert free_sl > 8 << 30 and free_ss > 2 << 30
    sizes = {}
    for gb in ("8", "4", "2"):
        monkeypatch.setenv("PDC_WORK_BUDGET_GB", gb)
        sizes[gb] = (lib.pdc_stringlength_work_bytes(n, n_per), lib.pdc_supersmoother_work_bytes(n, n_per),
                     lib.pdc_phase_work_bytes(3, n, n_per, 0, 0))
        assert max(sizes[gb]) <= int(gb) << 30, (gb, sizes[gb])
        assert sizes[gb][2] == sizes[gb][0]                  # the generic entry asks the same function
    assert sizes["2"][0] < sizes["4"][0] < sizes["8"][0] < free_sl
    assert sizes["2"][1] < free_ss
    # a curve that needs little is not touched by a budget it fits in
    monkeypatch.delenv("PDC_WORK_BUDGET_GB")
    small = lib.pdc_stringlength_work_bytes(5000, 100)
    monkeypatch.setenv("PDC_WORK_BUDGET_GB", "4")
    assert lib.pdc_stringlength_work_bytes(5000, 100) == small
    monkeypatch.delenv("PDC_WORK_BUDGET_GB")
    assert lib.pdc_stringlength_work_bytes(n, n_per) == free_sl
