"""In-kernel cycle stamps of k_diag_factor's phases at the C4 / C5 geometry (dev tool; needs a library whose rom_fem_kernels.hip
was compiled with -DROMHC_DIAGF_STAMPS: waves 0 and 513 print their stamps; 100 MHz s_memtime ticks are NOT what
__builtin_readcyclecounter returns here -- it counts shader clocks).
usage: python tools/dev/gpu_diagf_stamps.py path/to/libromhc_stamps.so [c4|c5]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd import _ffi
_ffi.load_library(os.path.abspath(sys.argv[1]))
cfg = sys.argv[2] if len(sys.argv) > 2 else "c4"
blocks, N, M = ((3, 3), 171, 1024) if cfg == "c4" else ((4, 4), 256, 4096)
ctx = _ffi.get_context(0)
fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
a = 10.0 ** np.random.default_rng(7).uniform(0, 2, size=(M, blocks[0] * blocks[1]))
ab = ctx.upload(a)
Y = ctx.alloc(M * fem.reduced_stride)
for rep in range(2):
    print("pass", rep, flush=True)
    fem.solve_reduced(ab, M, Y)
    ctx.solve_status()
