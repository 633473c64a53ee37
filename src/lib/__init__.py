"""Drop-in import path of the reference (``src.lib.X`` and, with ``src/`` on sys.path, ``lib.X``).

The implementation lives in ``romhighcontrast_amd.lib`` (HIP-backed); these modules only re-export it.
"""
