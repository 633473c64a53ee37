"""Repeated timings + host profile of pod_modes_factored at C2 size (dev probe)."""
import os, sys, time, cProfile, pstats, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd.lib import SolutionsManagers as SM
from romhighcontrast_amd import factored
import bench
logging.disable(logging.WARNING)
M = int(os.environ.get("M", "1024"))
sm = SM.SolutionsManagerFEM((2, 2), 128)
ctx, fem = sm._ctx, sm._fem
a = bench.workload_parameters("c2", (2, 2), M).reshape(M, -1)
Y = ctx.alloc(M * fem.reduced_stride)
fem.solve_reduced(ctx.upload(a), M, Y)
ctx.solve_status()
fs = factored.FactoredSnapshots(sm, Y, M)
for rep in range(6):
    ctx.synchronize(); t0 = time.perf_counter()
    factored.pod_modes_factored(fs, 50)
    ctx.synchronize(); print(f"rep {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable()
factored.pod_modes_factored(fs, 50); ctx.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
