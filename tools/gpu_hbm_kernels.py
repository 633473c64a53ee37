"""The HBM-bound kernels of the path on blocks of C2 / C4 size (dev tool, also the workload of the r02 PMC passes):
k_assemble_stencil (rom_assemble_batch), k_stencil_apply, k_h10_partial (H10norm and the greedy's residual norm),
k_sq_partial (l2norm).  Prints HIP-event time per launch and the ALGORITHMIC HBM rate (bytes each entry must move once)
against the 8 TB/s peak.  env: CFG=c2|c4, M (rows), REPS."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd import _ffi

cfg = os.environ.get("CFG", "c2")
NB, N = (2, 128) if cfg == "c2" else (3, 171)
M = int(os.environ.get("M", "1024"))
reps = int(os.environ.get("REPS", "5"))
ctx = _ffi.get_context(0)
fem = _ffi.Fem(ctx, NB, NB, N)
dim, nr, nc = fem.dim, fem.nr, fem.nc
a = 10.0 ** np.random.default_rng(1).uniform(0, 2, size=(M, NB * NB))
ab = ctx.upload(a)
U = ctx.alloc(M * dim)
fem.solve_batch(ab, M, U)
V = ctx.alloc(M * dim).copy_from(U, M * dim)
V.scale(1.0 + 1e-3)
Ma = min(M, 256)  # the three stencil arrays of `Ma` parameters
d, e, n = ctx.alloc(Ma * dim), ctx.alloc(Ma * nr * (nc - 1)), ctx.alloc(Ma * (nr - 1) * nc)
Y = ctx.alloc(M * dim)


def run():
    fem.assemble_batch(ab, Ma, d, e, n)
    fem.stencil_apply(U, M, Y)                         # unit coefficient (A_1 u)
    fem.stencil_apply(U, M, Y, a_one=a[0])             # block coefficients
    fem.h10norm(U, M)
    fem.h10norm(U, M, V=V)
    ctx.l2norm(U, 0, M, dim)


run()
ctx.synchronize()
ctx.profile(True)
for _ in range(reps):
    run()
rep = ctx.profile_report()
ctx.profile(False)
# algorithmic bytes per launch (each entry read or written once; SURVEY 8d: H10norm 16 B per entry incl. the partner)
alg = {"assemble_stencil": 24.0 * dim * Ma, "stencil_apply": 16.0 * dim * M, "h10norm": None, "l2norm": 8.0 * dim * M}
print(f"{cfg}: {NB}x{NB} blocks, N={N}, dim {dim}, {M} rows ({M * dim * 8 / 1e6:.0f} MB per block)")
for k, v in sorted(rep.items()):
    if v["launches"] == 0 or k.startswith("finish"):
        continue
    ms = v["total_ms"] / v["launches"]
    by = v["bytes"] / v["launches"]
    print(f"  {k:18s} {ms:8.4f} ms/launch  {by / ms * 1e-9:6.2f} TB/s of {by / 1e6:8.1f} MB accounted  = {by / ms * 1e-9 / 8.0:5.1%} of 8 TB/s"
          f"   ({v['launches']} launches)")
