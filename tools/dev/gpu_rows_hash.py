"""Hash of the rows of a few sweeps (dev tool): run it against two builds to see whether a kernel change kept the bits.
usage: python tools/dev/gpu_rows_hash.py [path/to/libromhc.so]"""
import sys, hashlib, os
sys.path.insert(0, ".")
import numpy as np
from romhighcontrast_amd import _ffi
if len(sys.argv) > 1:
    _ffi.load_library(os.path.abspath(sys.argv[1]))
ctx = _ffi.get_context(0)
for blocks, N, M in (((2, 2), 128, 300), ((2, 2), 64, 130), ((1, 2), 128, 70), ((3, 3), 171, 64), ((4, 4), 64, 40), ((2, 3), 40, 20)):
    a = 10.0 ** np.random.default_rng(N).uniform(0, 6, size=(M, blocks[0] * blocks[1]))
    fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
    U = ctx.alloc(M * fem.dim)
    fem.solve_batch(ctx.upload(a), M, U)
    h = hashlib.sha256(U.download().tobytes()).hexdigest()[:16]
    print(blocks, N, M, h)
