"""TEST INFRASTRUCTURE — loads the *unmodified* reference modules in the build container.

``/root/reference/src/periodicity/core.py:6`` imports xarray, which is not installed here,
so ``periodicity.spectral`` / ``periodicity.phase`` cannot be imported through the package.
Both files only use ``TSeries``/``FSeries`` from ``.core`` (``spectral.py:5``, ``phase.py:5``),
so this module registers a stand-in ``periodicity.core`` exposing the numpy-only containers of
``periodicity_amd.core`` and then executes the reference's two source files *where they lie*
(nothing is copied into the repo).  All arithmetic of the hot path then runs from the
reference's own source.

Only ``tests/`` and ``tests/golden/make_golden.py`` may import this; it is a no-op
(``available() == False``) on the GPU box where ``/root/reference`` does not exist.
"""
import importlib.util
import os
import sys
import types

REF_SRC = "/root/reference/src/periodicity"


def available():
    return os.path.isfile(os.path.join(REF_SRC, "spectral.py"))


def load():
    """Return ``(spectral, phase)`` — the reference modules, executed verbatim."""
    if not available():
        raise RuntimeError("reference sources are not present on this machine")
    if "periodicity.spectral" in sys.modules and "periodicity.phase" in sys.modules:
        return sys.modules["periodicity.spectral"], sys.modules["periodicity.phase"]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from periodicity_amd import core as _core

    pkg = types.ModuleType("periodicity")
    pkg.__path__ = []  # a package with no importable files of its own
    sys.modules["periodicity"] = pkg
    sys.modules["periodicity.core"] = _core
    mods = []
    for name in ("spectral", "phase"):
        spec = importlib.util.spec_from_file_location(
            f"periodicity.{name}", os.path.join(REF_SRC, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"periodicity.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
        mods.append(mod)
    return tuple(mods)
