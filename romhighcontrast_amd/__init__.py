"""romhighcontrast_amd -- MI355X-native snapshot + reduced-basis engine (hot path of ROMHighContrast).

Only the pieces the hot path needs: ``csrc/`` (HIP kernels + C-ABI, built into libromhc.so),
``_ffi`` (ctypes binding) and ``lib/`` (host-side mirror of the reference's src/lib API).
"""
__all__ = ["lib"]
