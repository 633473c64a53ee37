"""Stage timing of pod_modes as bench.py calls it (dev tool)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from romhighcontrast_amd.lib.SolutionsManagers import DeviceArray, SolutionsManagerFEM
from romhighcontrast_amd.lib import ReducedBasis as RB
ctx = _ffi.get_context(0)
M, N, r = 1024, 128, 50
sm = SolutionsManagerFEM((2, 2), N); dim = sm.vspace_dim
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 2, 2))
U = sm.generate_solutions_device(a)
X = ctx.alloc(M * dim)
for rep in range(3):
    X.copy_from(U.buf, M * dim); ctx.synchronize()
    t0 = time.perf_counter(); comps, sig = RB.pod_modes(ctx, DeviceArray(X, M, dim), r); ctx.synchronize()
    print(f"pod_modes: {1e3*(time.perf_counter()-t0):.1f} ms")
# wrap the context methods with timers
import collections
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        ctx.synchronize(); t = time.perf_counter(); out = f(*a, **k); ctx.synchronize()
        acc[name] += time.perf_counter() - t; cnt[name] += 1; return out
    setattr(obj, name, g)
for n in ("gram", "gemm_nt", "gemm_nn", "center_rows", "l2norm", "upload", "alloc"):
    wrap(ctx, n)
orig_dl = _ffi.Buffer.download
def dl(self, *a, **k):
    ctx.synchronize(); t = time.perf_counter(); out = orig_dl(self, *a, **k); acc["download"] += time.perf_counter() - t; cnt["download"] += 1; return out
_ffi.Buffer.download = dl
X.copy_from(U.buf, M * dim); ctx.synchronize()
t0 = time.perf_counter(); RB.pod_modes(ctx, DeviceArray(X, M, dim), r); ctx.synchronize(); tot = time.perf_counter() - t0
print(f"instrumented total {tot*1e3:.1f} ms; eigen iterations so far {RB._top_eigenpairs_device.total_iterations}; resolved {int((sig>0).sum())}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:12s} {v*1e3:8.2f} ms  calls {cnt[k]}")
print(f"  (python/host) {1e3*(tot-sum(acc.values())):.2f} ms")
