"""POD of a C2 / C3-size sweep from the interface vectors (factored.py) vs from materialised rows (dev tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import factored
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB

M = int(os.environ.get("M", "4096"))
nb, N = int(os.environ.get("NB", "2")), int(os.environ.get("N", "128"))  # C2/C3: NB=2 N=128; C5: NB=4 N=256
sm = SM.SolutionsManagerFEM((nb, nb), N)
fem, ctx = sm._fem, sm._ctx
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2 if nb == 2 else 3, size=(M, nb * nb))
ad = ctx.upload(a)
K = fem.reduced_stride
t0 = time.perf_counter()
em = factored.expansion_map(sm)
ctx.synchronize()
print(f"expansion map (S = B^T B, K = {K}): {time.perf_counter() - t0:.3f} s (once per FE space)")
Y = ctx.alloc(M * K)
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter()
    fem.solve_reduced(ad, M, Y); ctx.solve_status()
    t_red = time.perf_counter() - t0
print(f"reduced solves of {M} systems: {t_red * 1e3:.2f} ms")
fs = factored.FactoredSnapshots(sm, Y, M)
r = 50
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter()
    comps, sig = factored.pod_modes_factored(fs, r)
    ctx.synchronize(); t_f = time.perf_counter() - t0
F_pod = 2.0 * M * M * fem.dim + 2.0 * r * M * fem.dim + 10.0 * M ** 3
print(f"factored POD ({r} modes of {M} x {fem.dim}): {t_f * 1e3:.1f} ms = {F_pod / t_f * 1e-12:.1f} TFLOP/s in F_pod accounting; "
      f"sigma_1 {sig[0]:.6f}, resolved {int((sig > 0).sum())}")
if M * fem.dim * 8 < 40e9 and not os.environ.get("NO_ROWS"):
    U = fs.rows()
    for rep in range(2):
        X = SM.DeviceArray(ctx.alloc(M * fem.dim).copy_from(U.buf, M * fem.dim), M, fem.dim)
        ctx.synchronize(); t0 = time.perf_counter()
        comps_r, sig_r = RB.pod_modes(ctx, X, r)
        ctx.synchronize(); t_r = time.perf_counter() - t0
    print(f"rows POD: {t_r * 1e3:.1f} ms = {F_pod / t_r * 1e-12:.1f} TFLOP/s in F_pod accounting")
    big = sig_r > 1e-6 * sig_r[0]
    print(f"sigma rel diff (modes above 1e-6 sigma_1: {int(big.sum())}): {np.abs(sig[big] / sig_r[big] - 1).max():.2e}; "
          f"mode diff {np.abs(comps[big] - comps_r[big]).max():.2e}")
