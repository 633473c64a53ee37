"""Phase timing inside k_extend128 from in-kernel cycle stamps (dev tool; needs a library built with
`make -C romhighcontrast_amd/csrc EXTRA=-DROMHC_STAMPS`)."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
lib = ctx.lib
fem = _ffi.Fem(ctx, 2, 2, 128)
M = 1024
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 4))
ab, U = ctx.upload(a), ctx.alloc(M * fem.dim)
for _ in range(3):
    fem.solve_batch(ab, M, U)
ctx.synchronize()
n = 4064 * 5
buf = (C.c_ulonglong * n)()
assert lib.rom_debug_stamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 5).astype(np.int64)
t = t[(t[:, 4] > t[:, 0]) & (t[:, 0] > 0)]
d = np.diff(t[:, :5], axis=1)
names = ["entry -> first loads issued", "-> first barrier passed", "-> k loop done", "-> stores issued (end)"]
print(f"{len(t)} workgroups; lifetime mean {np.mean(t[:,4]-t[:,0]):.0f} cycles, median {np.median(t[:,4]-t[:,0]):.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:32s} mean {d[:, i].mean():8.0f}  median {np.median(d[:, i]):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
span = t[:, 4].max() - t[:, 0].min()
print(f"kernel span {span} cycles; sum of lifetimes / (512 slots x span) = {np.sum(t[:,4]-t[:,0]) / (512.0 * span):.2f}")

# ---- k_solve1 ----
if hasattr(lib, "rom_debug_stamps1"):
    n1 = 1024 * 6
    b1 = (C.c_ulonglong * n1)()
    assert lib.rom_debug_stamps1(b1, n1) == 0
    t1 = np.frombuffer(b1, dtype=np.uint64).reshape(-1, 6).astype(np.int64)
    t1 = t1[t1[:, 5] > t1[:, 0]]
    d1 = np.diff(t1, axis=1)
    print(f"k_solve1: {len(t1)} systems; lifetime mean {np.mean(t1[:,5]-t1[:,0]):.0f} cycles; first start -> last end {t1[:,5].max()-t1[:,0].min()} cycles")
    for i, nm in enumerate(["assembly", "rhs", "Cholesky", "back substitution", "coefficient blocks"]):
        print(f"  {nm:20s} mean {d1[:, i].mean():8.0f}  median {np.median(d1[:, i]):8.0f}  p90 {np.percentile(d1[:, i], 90):8.0f}")
