import sys
sys.path.insert(0, ".")
import numpy as np
from romhighcontrast_amd import _ffi
from oracle import rom_oracle as ro
ctx = _ffi.get_context(0)
for (nrb, ncb, N, M) in ((6, 6, 48, 5), (8, 8, 40, 3), (5, 7, 64, 3)):
    a = 10.0 ** np.random.default_rng(N).uniform(0, 3, size=(M, nrb, ncb))
    g = ro.Geometry((nrb, ncb), N)
    fem = _ffi.Fem(ctx, nrb, ncb, N)
    U = ctx.alloc(M * g.dim)
    fem.solve_batch(ctx.upload(a.reshape(M, -1)), M, U)
    Ug = U.download(shape=(M, g.dim))
    Uo = ro.generate_solutions(g, a, "lsqsparse")
    e = (ro.H10norm(g, Ug - Uo) / ro.H10norm(g, Uo)).max()
    print((nrb, ncb), N, "tiles", fem.n_tiles, "err", e, flush=True)
