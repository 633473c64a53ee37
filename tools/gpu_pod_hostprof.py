"""Where does the host time of pod_modes go?  (dev tool: cProfile over three calls at C2)"""
import sys, time, os, cProfile, pstats
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from romhighcontrast_amd.lib.SolutionsManagers import DeviceArray, SolutionsManagerFEM
from romhighcontrast_amd.lib import ReducedBasis as RB
ctx = _ffi.get_context(0)
M = int(os.environ.get("M", "1024")); N = int(os.environ.get("N", "128")); r = 50
sm = SolutionsManagerFEM((2, 2), N)
dim = sm.vspace_dim
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 2, 2))
U = sm.generate_solutions_device(a)
X2 = ctx.alloc(M * dim)
def run():
    X2.copy_from(U.buf, M * dim)
    ctx.synchronize()
    t = time.perf_counter()
    RB.pod_modes(ctx, DeviceArray(X2, M, dim), r, passes=int(os.environ.get("PASSES", "2")))
    ctx.synchronize()
    return time.perf_counter() - t
print("warm", run())
print("times", [round(run() * 1e3, 2) for _ in range(3)])
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
