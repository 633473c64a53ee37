import os, sys, time, logging
import numpy as np
sys.path.insert(0, os.getcwd())
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from romhighcontrast_amd import _ffi
logging.disable(logging.WARNING)
ctx = _ffi.get_context()
rng = np.random.default_rng(3)
for M, dim, n in ((1500, 2500, 40), (2048, 3000, 30), (2300, 2600, 20)):
    Xh = rng.standard_normal((M, dim))
    sv = np.linalg.svd(Xh, compute_uv=False)
    t0 = time.perf_counter()
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), M, dim), n, center=False)
    dt = time.perf_counter() - t0
    print(M, dim, n, f"{dt:.2f} s", "rel err", np.abs(sig / sv[:n] - 1).max(), {k: RB.pod_modes.last_info[k] for k in ("gram_passes", "subspace_iterations", "stop_reason")})
