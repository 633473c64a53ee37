import sys
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
for blocks, N in (((2, 2), 128), ((3, 3), 171), ((4, 4), 256), ((2, 2), 64), ((4, 4), 64), ((8, 8), 32)):
    fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
