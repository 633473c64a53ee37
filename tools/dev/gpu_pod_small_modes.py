"""Dev probe: per-mode accuracy of rom_pod against LAPACK's SVD on the scenario of test_pod_small_modes_vs_lapack."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
sm = SM.SolutionsManagerFEM((2, 2), 32)
ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
M, n = 200, 30
a = 10.0 ** np.random.default_rng(4).uniform(0, 2, size=(M, 2, 2))
Ud = sm.generate_solutions_device(a)
Uh = Ud.numpy()
_, sv, Vt = np.linalg.svd(Uh - Uh.mean(axis=0), full_matrices=False)
print("sv/sv0:", " ".join(f"{v:.1e}" for v in sv[:n] / sv[0]))
X = ctx.alloc(M * dim).copy_from(Ud.buf, M * dim)
comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), n)
print("info", RB.pod_modes.last_info)
print("sig rel err:", " ".join(f"{abs(s / v - 1):.1e}" for s, v in zip(sig, sv[:n])))
real = sv[:n] > 1e-11 * sv[0]
for k in range(2, int(real.sum()) + 1):
    P1, P2 = comps[:k].T @ comps[:k], Vt[:k].T @ Vt[:k]
    print(k, f"{sv[k-1]/sv[0]:.1e}", f"projector diff {np.abs(P1 - P2).max():.2e}", f"mode angle {np.sqrt(max(0, 1 - (comps[k-1] @ Vt[k-1]) ** 2)):.2e}")
