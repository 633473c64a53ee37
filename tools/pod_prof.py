"""Where the time of pod_modes goes (dev probe): per-kernel HIP-event totals vs wall, C3-size block by default."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
import bench
M = int(os.environ.get("M", "8192"))
sm = SM.SolutionsManagerFEM((2, 2), 128)
ctx = sm._ctx
dim = sm.vspace_dim
a = bench.workload_parameters("c2", (2, 2), M)
U = sm.generate_solutions_device(a)
X = ctx.alloc(M * dim)
for rep in range(2):
    X.copy_from(U.buf, M * dim)
    ctx.synchronize()
    ctx.profile_reset(); ctx.profile(True)
    t0 = time.perf_counter()
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), 50)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    rep_ = ctx.profile_report()
tot = sum(v["total_ms"] for v in rep_.values())
print(f"wall {dt*1e3:.1f} ms, kernels {tot:.1f} ms, info {RB.pod_modes.last_info}")
for k, v in sorted(rep_.items(), key=lambda kv: -kv[1]["total_ms"])[:int(os.environ.get("TOP", "12"))]:
    print(f"  {k:34s} {v['total_ms']:8.2f} ms  launches {v['launches']:5d}  {v['flops']/max(v['total_ms'],1e-9)*1e-9:7.2f} TFLOP/s")
