"""POD stage timings (dev tool)."""
import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from romhighcontrast_amd.lib.SolutionsManagers import DeviceArray, SolutionsManagerFEM
from romhighcontrast_amd.lib import ReducedBasis as RB
ctx = _ffi.get_context(0)
M = int(os.environ.get("M", "1024")); N = int(os.environ.get("N", "128")); r = 50
sm = SolutionsManagerFEM((2, 2), N)
dim = sm.vspace_dim
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 2, 2))
U = sm.generate_solutions_device(a)
def T(f):
    ctx.synchronize(); t = time.perf_counter(); out = f(); ctx.synchronize(); return out, time.perf_counter() - t
X = ctx.alloc(M * dim).copy_from(U.buf, M * dim)
_, t = T(lambda: ctx.center_rows(X, M, dim, ctx.alloc(dim))); print(f"center {t*1e3:.2f} ms")
G = ctx.alloc(M * M)
_, t = T(lambda: ctx.gram(M, dim, X, 0, dim, G, 0, M)); _, t = T(lambda: ctx.gram(M, dim, X, 0, dim, G, 0, M))
print(f"gram (lower+mirror) {t*1e3:.2f} ms -> {M*(M+64)*dim/t*1e-12:.2f} TFLOP/s (computed flops)")
G2 = ctx.alloc(M * M)
_, t = T(lambda: ctx.gemm_nt(M, M, dim, X, 0, dim, X, 0, dim, G2, 0, M)); _, t = T(lambda: ctx.gemm_nt(M, M, dim, X, 0, dim, X, 0, dim, G2, 0, M))
print(f"gemm_nt full {t*1e3:.2f} ms -> {2*M*M*dim/t*1e-12:.2f} TFLOP/s")
g1 = G.download(M * M, shape=(M, M)); g2 = G2.download(M * M, shape=(M, M))
print("gram vs gemm_nt max rel diff", np.abs(g1 - g2).max() / np.abs(g2).max(), "symmetric:", bool((g1 == g1.T).all()))
(lam, W), t = T(lambda: RB._top_eigenpairs_device(ctx, G, M, r)); print(f"eigs {t*1e3:.2f} ms, {RB._top_eigenpairs_device.last_iterations} iterations")
Gh = G.download(M * M, shape=(M, M)); t0 = time.perf_counter(); lr = np.linalg.eigvalsh(Gh)[::-1]; print(f"host eigvalsh {1e3*(time.perf_counter()-t0):.1f} ms; resolvable {int((lr[:r] > 1e-13*lr[0]).sum())}, their max rel err {np.max((np.abs(lam-lr[:r])/lr[:r])[lr[:r] > 1e-13*lr[0]]):.2e}")
V = ctx.alloc(r * dim)
_, t = T(lambda: ctx.gemm_nn(r, dim, M, W.buf, 0, M, X, 0, dim, V, 0, dim)); print(f"lift {t*1e3:.2f} ms")
_, t = T(lambda: RB._orthonormalize_device(ctx, DeviceArray(V, r, dim))); print(f"orth {r} x dim {t*1e3:.2f} ms")
X2 = ctx.alloc(M * dim).copy_from(U.buf, M * dim)
(_, sig), t = T(lambda: RB.pod_modes(ctx, DeviceArray(X2, M, dim), r, passes=1)); print(f"pod_modes total {t*1e3:.1f} ms  sigma[0]={sig[0]:.4e} sigma[{r-1}]={sig[-1]:.3e}")
ctx.profile(True)
X2.copy_from(U.buf, M * dim)
RB.pod_modes(ctx, DeviceArray(X2, M, dim), r, passes=1)
for k, v in sorted(ctx.profile_report().items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
    print(f"  {k:16s} {v['total_ms']:8.3f} ms launches {v['launches']:5d} {v['flops']/v['total_ms']*1e-9 if v['total_ms'] else 0:7.2f} TFLOP/s")
