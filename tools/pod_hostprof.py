"""cProfile of pod_modes on the host side (dev probe).  env: M."""
import os, sys, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
import bench
M = int(os.environ.get("M", "1024"))
sm = SM.SolutionsManagerFEM((2, 2), 128)
ctx = sm._ctx
dim = sm.vspace_dim
a = bench.workload_parameters("c2", (2, 2), M)
U = sm.generate_solutions_device(a)
X = ctx.alloc(M * dim)
import logging; logging.disable(logging.WARNING)
for rep in range(2):
    X.copy_from(U.buf, M * dim)
    ctx.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), 50)
    ctx.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("TOP", "28")))
