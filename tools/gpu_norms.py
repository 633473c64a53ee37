"""HBM rates of the norm kernels on a snapshot block (dev tool): env NB, N, M."""
import sys, os
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
NB = int(os.environ.get("NB", "2")); N = int(os.environ.get("N", "128")); M = int(os.environ.get("M", "1024"))
fem = _ffi.Fem(ctx, NB, NB, N)
a = 10.0 ** np.random.default_rng(1).uniform(0, 2, size=(M, NB * NB))
U = ctx.alloc(M * fem.dim)
fem.solve_batch(ctx.upload(a), M, U)
V = ctx.alloc(M * fem.dim).copy_from(U, M * fem.dim)
V.scale(1.0 + 1e-3)
for rep in range(2):
    h = fem.h10norm(U, M)
    l = ctx.l2norm(U, 0, M, fem.dim)
    d = fem.h10norm(U, M, V=V) if "V" in fem.h10norm.__code__.co_varnames else None
ctx.profile(True)
for rep in range(5):
    fem.h10norm(U, M)
    ctx.l2norm(U, 0, M, fem.dim)
    if d is not None:
        fem.h10norm(U, M, V=V)
for k, v in sorted(ctx.profile_report().items()):
    print(f"{k:12s} {v['total_ms']/v['launches']:.4f} ms/launch  {v['bytes']/v['total_ms']*1e-9:.2f} TB/s (accounted bytes)  launches {v['launches']}")
print("h10[0:3]", h[:3], "l2[0:3]", l[:3])
