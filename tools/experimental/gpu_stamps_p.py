"""Per-chunk cycle stamps of k_extend_p (workgroup 3, waves 0 and 3; debug build: make EXTRA=-DROMHC_STAMPS)."""
import os
import sys
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ROMHC_EXT_P", "1")
from romhighcontrast_amd import _ffi

ctx = _ffi.get_context(0)
lib = ctx.lib
NB, N, M = int(os.environ.get("NB", "2")), int(os.environ.get("N", "128")), int(os.environ.get("M", "1024"))
fem = _ffi.Fem(ctx, NB, NB, N)
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, NB * NB))
ab, U = ctx.upload(a), ctx.alloc(M * fem.dim)
Y = ctx.alloc(M * fem.reduced_stride)
fem.solve_reduced(ab, M, Y)
for _ in range(3):
    fem.expand(ab, M, Y, U)
ctx.synchronize()
lib.rom_debug_stamps_p_clear()
ctx.timer_start()
fem.expand(ab, M, Y, U)
ms = ctx.timer_stop()
n = 2048 * 6
buf = (C.c_ulonglong * n)()
assert lib.rom_debug_stamps_p(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 6).astype(np.int64)
print(f"launch {ms * 1e3:.1f} us")
for wave, base in (("wave 0", 0), ("wave 3", 1024)):
    w = t[base:base + 1024]
    w = w[w[:, 0] > 0]
    if not len(w):
        continue
    d = np.diff(w[:, :5], axis=1)
    period = np.diff(w[:, 0])
    print(f"{wave}: {len(w)} chunks; chunk period mean {period.mean():.0f} cycles (median {np.median(period):.0f}); span {w[-1, 4] - w[0, 0]} cycles")
    for i, nm in enumerate(["k-steps 0,1 (+ stores)", "wait + barrier", "7 loads issued", "k-steps 2,3 (+ stores)"]):
        print(f"   {nm:42s} mean {d[:, i].mean():7.0f}  median {np.median(d[:, i]):7.0f}  p90 {np.percentile(d[:, i], 90):7.0f}")
    gap = w[1:, 0] - w[:-1, 4]
    print(f"   {'end of chunk -> start of next':42s} mean {gap.mean():7.0f}  median {np.median(gap):7.0f}  p90 {np.percentile(gap, 90):7.0f}")
    first = (w[1:, 5] & 0xff) == 0
    if first.any():
        print(f"   ... of which at tile boundaries: mean {gap[first].mean():.0f}, inside tiles: mean {gap[~first].mean():.0f}")
