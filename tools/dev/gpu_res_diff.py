import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context()
blocks, N, M = (2, 2), 128, int(sys.argv[1]) if len(sys.argv) > 1 else 130
a = 10.0 ** np.random.default_rng(N).uniform(0, 3, size=(M, 4))
ab = ctx.upload(a)
out = {}
for name, env in (("ref", {"ROMHC_NO_EXT_RES": "1"}), ("res", {})):
    os.environ.pop("ROMHC_NO_EXT_RES", None)
    os.environ.update(env)
    fem = _ffi.Fem(ctx, 2, 2, N)
    U = ctx.alloc(M * fem.dim); U.fill(float("nan"))
    fem.solve_batch(ab, M, U)
    out[name] = U.download(shape=(M, fem.dim))
d = out["res"] != out["ref"]
d |= np.isnan(out["res"]) != np.isnan(out["ref"])
print("mismatching entries", d.sum(), "rows", np.flatnonzero(d.any(axis=1))[:20], "nan in res", np.isnan(out["res"]).sum())
if d.any():
    r = np.flatnonzero(d.any(axis=1))[0]
    cols = np.flatnonzero(d[r])
    nc = 255
    print("row", r, "n cols", len(cols), "mesh (r,c) of first:", [(c // nc, c % nc) for c in cols[:12]], "last:", [(c // nc, c % nc) for c in cols[-4:]])
    print("values res/ref", out["res"][r, cols[:4]], out["ref"][r, cols[:4]])
    mr = sorted(set(c // nc for c in cols)); print("mesh rows affected:", mr[:20], len(mr))
