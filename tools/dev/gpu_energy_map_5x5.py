import sys, time
sys.path.insert(0, ".")
import numpy as np
from romhighcontrast_amd import _ffi, factored
from romhighcontrast_amd.lib import SolutionsManagers as SM
sm = SM.SolutionsManagerFEM((5, 5), 64)
ctx, fem = sm._ctx, sm._fem
print("5x5/N=64: compact stride", fem.compact_stride, "tiles", fem.n_tiles, "linear", fem.expansion_is_linear)
t0 = time.perf_counter(); print("energy map ranks", fem.energy_map(7), f"{(time.perf_counter() - t0) * 1e3:.1f} ms")
a = 10.0 ** np.random.default_rng(1).uniform(0, 3, size=(64, 5, 5))
Ud = sm.generate_solutions_device(a)
h1 = sm.H10norm(Ud)
print("h10 factored vs rows", np.abs(factored.h10norm_factored(Ud.factored) / h1 - 1).max())
