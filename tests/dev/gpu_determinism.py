"""Are repeated sweeps bit-identical?  (dev probe: golden g4 geometries, fresh FE spaces, interleaved other work)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from oracle import rom_oracle as ro
ctx = _ffi.get_context(0)
z = np.load("tests/golden/g4_contrast.npz")
bad = 0
for name in ("b22", "b33", "b44"):
    blocks, N = tuple(int(x) for x in z[f"{name}_blocks"]), int(z[f"{name}_N"])
    a = np.asarray(z[f"{name}_a"], dtype=np.float64).reshape(len(z[f"{name}_a"]), -1)
    M = len(a)
    g = ro.Geometry(blocks, N)
    ref = None
    for rep in range(40):
        if rep % 5 == 0:
            fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
            # other work in between: a different geometry churns the allocator and the caches
            other = _ffi.Fem(ctx, 2, 2, 16 + rep)
            Uo = ctx.alloc(8 * other.dim)
            other.solve_batch(ctx.upload(np.full((8, 4), 2.0)), 8, Uo)
        U = ctx.alloc(M * fem.dim)
        U.fill(float("nan"))
        fem.solve_batch(ctx.upload(a), M, U)
        out = U.download(shape=(M, fem.dim))
        if ref is None:
            ref = out
            Uref = z[f"{name}_U"]
            e = ro.H10norm(g, out - Uref) / ro.H10norm(g, Uref)
            print(name, blocks, N, "M", M, "rel H10 err vs golden", np.array2string(e, precision=2))
        elif not np.array_equal(out, ref):
            bad += 1
            rows = np.flatnonzero((out != ref).any(axis=1))
            e = ro.H10norm(g, out - Uref) / ro.H10norm(g, Uref)
            print(f"  {name} rep {rep}: differs in rows {rows}; nan {np.isnan(out).sum()}; max abs diff {np.nanmax(np.abs(out - ref)):.3e}; err vs golden {np.array2string(e[rows], precision=2)}")
print("deviating repetitions:", bad)
