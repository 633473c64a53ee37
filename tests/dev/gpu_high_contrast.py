import sys
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from oracle import rom_oracle as ro
ctx = _ffi.get_context(0)
INF = 1e10
for blocks, N in (((3, 3), 40), ((2, 2), 96), ((4, 4), 24)):
    g = ro.Geometry(blocks, N)
    k = blocks[0] * blocks[1]
    rows = [np.ones(k)]
    for j in range(k):
        r = np.ones(k); r[j] = INF; rows.append(r)
    r = np.full(k, INF); rows.append(r)
    r = np.full(k, INF); r[0] = 1; rows.append(r)
    rng = np.random.default_rng(5)
    for _ in range(6):
        rows.append(10.0 ** rng.uniform(0, 10, k))
    for _ in range(4):
        rows.append(rng.choice([1.0, INF, 1e5], size=k))
    a = np.array(rows)
    M = len(a)
    fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
    U = ctx.alloc(M * fem.dim)
    fem.solve_batch(ctx.upload(a), M, U)
    Ug = U.download(shape=(M, fem.dim))
    Uo = ro.generate_solutions(g, a.reshape((M,) + blocks), "lsqsparse")
    e = ro.H10norm(g, Ug - Uo) / ro.H10norm(g, Uo)
    print(blocks, N, "max rel H10 err per row:", np.array2string(e, precision=1))
