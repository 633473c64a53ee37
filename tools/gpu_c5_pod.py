"""BASELINE config C5 end to end (dev tool): 4096-snapshot sweep on the 1024 x 1024 grid (4x4 blocks, N=256),
POD of the 4096 x 1 046 529 block via the MFMA Gram matrix, Galerkin projection error of the leading modes."""
import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from romhighcontrast_amd.lib.SolutionsManagers import SolutionsManagerFEM, DeviceArray
from romhighcontrast_amd.lib import ReducedBasis as RB
ctx = _ffi.get_context(0)
M, r = int(os.environ.get("M", "4096")), 50
sm = SolutionsManagerFEM((4, 4), 256); dim = sm.vspace_dim
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 3, size=(M, 4, 4))
def T(f):
    ctx.synchronize(); t = time.perf_counter(); out = f(); ctx.synchronize(); return out, time.perf_counter() - t
U, t = T(lambda: sm.generate_solutions_device(a)); print(f"sweep M={M}: {t:.2f} s -> {M/t:.0f} solves/s", flush=True)
h1, t = T(lambda: sm.H10norm(U)); print(f"H10norm: {t*1e3:.1f} ms -> {8.0*M*dim/t*1e-9:.0f} GB/s", flush=True)
X = ctx.alloc(M * dim).copy_from(U.buf, M * dim)
G = ctx.alloc(M * M)
_, t = T(lambda: ctx.center_rows(X, M, dim, ctx.alloc(dim))); print(f"centre: {t*1e3:.1f} ms", flush=True)
_, t = T(lambda: ctx.gram(M, dim, X, 0, dim, G, 0, M)); print(f"Gram (lower tiles + mirror): {t:.3f} s -> {M*(M+64)*dim/t*1e-12:.1f} TFLOP/s computed, {2.0*M*M*dim/t*1e-12:.1f} TFLOP/s in 2 M^2 D accounting", flush=True)
X.copy_from(U.buf, M * dim)
(comps, sig), t = T(lambda: RB.pod_modes(ctx, DeviceArray(X, M, dim), r, passes=1))
f_pod = 2.0 * M * M * dim + 2.0 * r * M * dim + 10.0 * M ** 3
print(f"pod_modes (1 pass, {int((sig>0).sum())} modes): {t:.3f} s -> {f_pod/t*1e-12:.1f} TFLOP/s in F_pod accounting; sigma_1={sig[0]:.3e} sigma_r={sig[sig>0][-1]:.3e}", flush=True)
# Galerkin ROM error of the leading modes on the first 256 parameters
nb = int((sig > 0).sum())
C = comps[:nb]
approx = sm.generate_fm_solutions_device(a[:256], C)
err = sm.H10norm_diff(approx, DeviceArray(U.buf, 256, dim)) / h1[:256]
print(f"Galerkin ROM with {nb} POD modes: max rel H10 error over 256 parameters {err.max():.2e}, median {np.median(err):.2e}")
