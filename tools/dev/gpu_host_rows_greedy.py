"""The reference's call pattern at C4: solutions = sm.generate_solutions(a) (host), build(n, sm, solutions, a, h1) -- with the
array the manager returned (interface-vector route after the bit-for-bit check) and with a copy of it (row route) (dev probe)."""
import os, sys, time, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
import bench
logging.disable(logging.WARNING)
sm = SM.SolutionsManagerFEM((3, 3), 171)
a = bench.workload_parameters("c4", (3, 3), 1024)
t0 = time.perf_counter(); U = sm.generate_solutions(a); print(f"generate_solutions (host): {time.perf_counter() - t0:.3f} s")
U = sm.generate_solutions(a)
h1 = sm.H10norm(U)
Uc = U.copy()
for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
    for name, arr in (("returned array", U), ("copy", Uc)):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            rb = RB.ReducedBasisGreedy(mode).build(50, sm, arr, a, h1)
            ts.append(time.perf_counter() - t0)
        print(f"{mode:10s} {name:15s}: {min(ts) * 1e3:7.1f} ms (of which the upload of 2.1 GB is common to both)")
