"""The 320-mode request of tests/test_gpu_parity.py::test_pod_slowly_decaying_spectrum_many_modes, per-mode errors (dev probe)."""
import os, sys, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from romhighcontrast_amd import _ffi
logging.disable(logging.WARNING)
ctx = _ffi.get_context()
rng = np.random.default_rng(11)
M, dim, n = 512, 3000, 320
Q1, _ = np.linalg.qr(rng.standard_normal((M, M)))
Q2, _ = np.linalg.qr(rng.standard_normal((dim, M)))
s = 10.0 ** (-np.arange(M) / 27.0)
Xh = (Q1 * s) @ Q2.T
sv = np.linalg.svd(Xh, compute_uv=False)
comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), M, dim), n, center=False)
print(RB.pod_modes.last_info)
err = np.abs(sig - sv[:n])
for i in list(range(0, n, 20)) + list(range(n - 12, n)):
    print(f"{i:4d} sv {sv[i]:.3e} abs err {err[i]:.2e} = {err[i] / 1.1e-16:.0f} eps sigma_1, rel {err[i] / sv[i]:.1e}")
print("worst beyond 1e-14:", (np.maximum(err - 1e-14, 0) / sv[:n]).max(), "at", int((np.maximum(err - 1e-14, 0) / sv[:n]).argmax()))
