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
