"""Wall-time segments of pod_modes at C2 (dev tool): wraps the context methods and the helpers with synchronising timers."""
import sys, time, os, collections
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
X2 = ctx.alloc(M * dim)
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        ctx.synchronize(); t = time.perf_counter()
        out = f(*a, **k)
        ctx.synchronize(); acc[label or name] += time.perf_counter() - t; cnt[label or name] += 1
        return out
    setattr(obj, name, g)
for nm in ("center_rows", "gram", "gemm_nt", "gemm_nn", "l2norm", "upload"):
    wrap(_ffi.Context, nm)
wrap(_ffi.Buffer, "download"); wrap(_ffi.Buffer, "scale"); wrap(_ffi.Buffer, "copy_from")
wrap(np.linalg, "eigh", "host eigh"); wrap(np.linalg, "cholesky", "host chol"); wrap(np.linalg, "inv", "host inv")
tot = 0
for rep in range(4):
    X2.copy_from(U.buf, M * dim)
    if rep == 1:
        acc.clear(); cnt.clear(); tot = 0
    ctx.synchronize(); t = time.perf_counter()
    RB.pod_modes(ctx, DeviceArray(X2, M, dim), r)
    ctx.synchronize(); tot += time.perf_counter() - t
print(f"pod_modes (with the timers' syncs): {tot/3*1e3:.2f} ms per call")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:14s} {v/3*1e3:7.3f} ms  ({cnt[k]//3} calls)")
print(f"  unaccounted    {(tot - sum(v for k, v in acc.items() if k != 'copy_from' or True))/3*1e3:7.3f} ms")
