"""Random geometries vs the oracle (dev tool): catches layout / symbolic-factorisation corner cases."""
import sys
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from oracle import rom_oracle as ro

ctx = _ffi.get_context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    nrb, ncb = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    N = int(rng.choice([2, 3, 4, 5, 7, 9, 12, 16, 17, 24, 31, 33, 40, 48, 64, 65, 70, 97, 100, 129, 130]))
    if nrb * ncb * N * N > 70000:
        nrb, ncb = min(nrb, 2), min(ncb, 2)
    if nrb * ncb * N * N > 70000:
        N = max(2, int((70000 / (nrb * ncb)) ** 0.5))
    M = int(rng.choice([1, 2, 5, 130, 200]))  # >= 128 systems: the wide-tile extension kernel
    a = 10.0 ** rng.uniform(0, rng.choice([1, 3, 6]), size=(M, nrb, ncb))
    g = ro.Geometry((nrb, ncb), N)
    fem = _ffi.Fem(ctx, nrb, ncb, N)
    U = ctx.alloc(M * g.dim)
    fem.solve_batch(ctx.upload(a.reshape(M, -1)), M, U)
    Ug = U.download(shape=(M, g.dim))
    Uo = ro.generate_solutions(g, a, "lsqsparse")
    e = (ro.H10norm(g, Ug - Uo) / ro.H10norm(g, Uo)).max()
    # two-stage path must be bit-identical
    Y = ctx.alloc(max(M * fem.reduced_stride, 1))
    fem.solve_reduced(ctx.upload(a.reshape(M, -1)), M, Y)
    U2 = ctx.alloc(M * g.dim)
    fem.expand(ctx.upload(a.reshape(M, -1)), M, Y, U2)
    ctx.solve_status()
    same = np.array_equal(U2.download(shape=(M, g.dim)), Ug)
    worst = max(worst, e)
    flag = "" if (e < 1e-11 and same) else "   <-- CHECK"
    print(f"{trial:3d} blocks=({nrb},{ncb}) N={N:3d} M={M} tiles={fem.n_tiles:3d} linear={int(fem.expansion_is_linear)} err {e:.2e} two-stage identical {same}{flag}", flush=True)
print("worst", worst)
