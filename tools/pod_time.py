"""Wall time of pod_modes on the C3-size block, several calls (dev probe).  env: M, REPS."""
import os, sys, time, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
import bench
logging.disable(logging.WARNING)
M = int(os.environ.get("M", "8192"))
sm = SM.SolutionsManagerFEM((2, 2), 128)
ctx = sm._ctx
dim = sm.vspace_dim
U = sm.generate_solutions_device(bench.workload_parameters("c2", (2, 2), M))
X = ctx.alloc(M * dim)
ts = []
for rep in range(int(os.environ.get("REPS", "6"))):
    X.copy_from(U.buf, M * dim)
    ctx.synchronize()
    t0 = time.perf_counter()
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), 50)
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
print("pod_modes walls (ms):", " ".join(f"{t * 1e3:.1f}" for t in ts))
