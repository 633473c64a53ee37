"""Fused sweep vs the two-stage sweep (reduced solve, then expansion) that the multi-GPU step uses (dev probe)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
N, M, K = 128, 1024, 50
fem = _ffi.Fem(ctx, 2, 2, N)
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 4))
ab, U = ctx.upload(a), ctx.alloc(M * fem.dim)
Y = [ctx.alloc(M * fem.reduced_stride) for _ in range(2)]
def fused():
    fem.solve_batch(ab, M, U, wait=False)
def two_stage(k=[0]):
    y = Y[k[0] & 1]; k[0] += 1
    fem.solve_reduced(ab, M, y)
    fem.expand(ab, M, y, U)
for name, step in (("fused", fused), ("two-stage", two_stage)):
    for _ in range(5):
        step()
    ctx.solve_status()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        ctx.solve_status()
        best = min(best, time.perf_counter() - t0)
    print(f"{name:10s} {best / K * 1e3:.4f} ms/step")
    ctx.profile(True)
    for _ in range(5):
        step()
    rep = ctx.profile_report()
    ctx.profile(False)
    print("   " + ", ".join(f"{k} {v['total_ms'] / 5:.4f}" for k, v in sorted(rep.items())))
