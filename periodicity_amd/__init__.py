"""MI355X-native trial-frequency scans (GLS / StringLength / PDM).

Like the reference package (``/root/reference/src/periodicity/__init__.py:1-2``) this
module re-exports nothing: import by submodule, e.g. ``periodicity_amd.spectral.GLS``.
"""
name = "periodicity_amd"
__version__ = "0.1.0"
