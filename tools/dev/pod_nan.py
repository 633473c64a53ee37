import os, sys, logging, time
import numpy as np
sys.path.insert(0, os.getcwd())
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from romhighcontrast_amd import _ffi
logging.disable(logging.WARNING)
ctx = _ffi.get_context()
rng = np.random.default_rng(0)
for name, mut in (("NaN entry", lambda X: X.__setitem__((3, 7), np.nan)), ("Inf entry", lambda X: X.__setitem__((5, 11), np.inf)), ("all zeros", lambda X: X.fill(0.0)), ("huge values", lambda X: X.__imul__(1e200)), ("tiny values", lambda X: X.__imul__(1e-200))):
    for center in (True, False):
        X = rng.standard_normal((200, 3000))
        mut(X)
        t0 = time.perf_counter()
        try:
            comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(X), 200, 3000), 20, center=center)
            print(name, "center", center, f"{time.perf_counter() - t0:.3f} s", "sig[:3]", sig[:3], RB.pod_modes.last_info["stop_reason"], "finite comps", bool(np.isfinite(comps).all()))
        except Exception as e:
            print(name, "center", center, "raised", type(e).__name__, str(e)[:100])
