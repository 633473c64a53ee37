"""Per-mode accuracy of pod_modes at the C2 block against LAPACK (dev probe).  env: M (1024), N (32)."""
import os, sys, time, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
import bench
logging.disable(logging.WARNING)
M = int(os.environ.get("M", "1024"))
n = int(os.environ.get("N", "32"))
sm = SM.SolutionsManagerFEM((2, 2), 128)
ctx = sm._ctx
dim = sm.vspace_dim
U = sm.generate_solutions_device(bench.workload_parameters("c2", (2, 2), M))
Uh = U.numpy()
ref = os.environ.get("REF", "gpurun_out/pod_angles_ref.npz")
_, sv, Vt = np.linalg.svd(Uh - Uh.mean(axis=0), full_matrices=False)
Vt = Vt[:n]
X = ctx.alloc(M * dim)
for rep in range(3):
    X.copy_from(U.buf, M * dim)
    ctx.synchronize()
    t0 = time.perf_counter()
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), n)
    ctx.synchronize()
    t = time.perf_counter() - t0
print("wall ms", t * 1e3, RB.pod_modes.last_info)
print("orthonormality defect", np.abs(comps @ comps.T - np.eye(n)).max())
for i in range(n):
    # angle of mode i to the LAPACK subspace of the modes with sigma within a factor 3 (close values rotate freely)
    grp = (sv[:n] < 3 * sv[i]) & (sv[:n] > sv[i] / 3)
    r = comps[i] - (comps[i] @ Vt[grp].T) @ Vt[grp]
    print(f"{i:3d} sigma/s1 lapack {sv[i] / sv[0]:.3e} ours {sig[i] / sv[0]:.3e} rel {abs(sig[i] / sv[i] - 1):.1e} angle {np.linalg.norm(r):.2e}  noise-bound {1e-16 * sv[0] / sv[i]:.1e}")
