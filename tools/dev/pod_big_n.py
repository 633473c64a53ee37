"""Requests of 100-300 modes: wall time of rom_pod on a slowly decaying synthetic block and on the C2 block (dev probe)."""
import os, sys, time, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from romhighcontrast_amd import _ffi
import bench
logging.disable(logging.WARNING)
ctx = _ffi.get_context()
rng = np.random.default_rng(5)
M, dim = 1024, 8000
Q1, _ = np.linalg.qr(rng.standard_normal((M, M)))
Q2, _ = np.linalg.qr(rng.standard_normal((dim, M)))
s = 10.0 ** (-np.arange(M) / 40.0)
Xh = (Q1 * s) @ Q2.T
for n in (50, 100, 200, 300):
    for rep in range(2):
        X = ctx.upload(Xh)
        ctx.synchronize()
        t0 = time.perf_counter()
        comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), n, center=False)
        dt = time.perf_counter() - t0
    print(f"slow decay (a decade per 40 modes) 1024 x 8000, n = {n}: {dt * 1e3:.1f} ms, sv rel err {np.abs(sig / s[:n] - 1).max():.1e}", {k: RB.pod_modes.last_info[k] for k in ("gram_passes", "sketch_passes", "subspace_iterations", "stop_reason")})
sm = SM.SolutionsManagerFEM((2, 2), 128)
U = sm.generate_solutions_device(bench.workload_parameters("c2", (2, 2), 1024))
X = sm._ctx.alloc(1024 * sm.vspace_dim)
for n in (100, 200):
    for rep in range(2):
        X.copy_from(U.buf, 1024 * sm.vspace_dim)
        sm._ctx.synchronize()
        t0 = time.perf_counter()
        comps, sig = RB.pod_modes(sm._ctx, SM.DeviceArray(X, 1024, sm.vspace_dim), n)
        dt = time.perf_counter() - t0
    print(f"C2 block, n = {n}: {dt * 1e3:.1f} ms", {k: RB.pod_modes.last_info[k] for k in ("resolved_modes", "completed_modes", "sketch_passes")})
