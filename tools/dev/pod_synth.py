"""How many orders of magnitude one sketch pass of rom_pod resolves: a block with a known SVD -- 16 modes from 1 to 1e-5
(the Gram stage's), then one mode per 10^(-1/PER) from 1e-6 down to 1e-12, on a noise floor of 1e-15 (dev probe)."""
import os, sys, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from romhighcontrast_amd import _ffi
logging.disable(logging.WARNING)
ctx = _ffi.get_context()
rng = np.random.default_rng(5)
M, dim = 512, 6000
per = float(os.environ.get("PER", "1"))
lead = 10.0 ** -np.linspace(0, 5, 16)
deep = 10.0 ** -np.arange(6, 12.0001, 1.0 / per)
if os.environ.get("DEEP8"):   # eight orders below the reach of the Gram stage, down to the floor
    deep = np.concatenate([0.9e-5 * 10.0 ** -np.arange(0, 7.0001, 1.0 / per), [2e-13]])
s = np.concatenate([lead, deep])
if os.environ.get("FULL13"):   # one mode per 10^(-1/PER) from 1 down to 2e-13
    s = np.concatenate([10.0 ** -np.arange(0, 12.5, 1.0 / per), [2e-13]])
if os.environ.get("SLOW"):   # one decade per SLOW modes, 50 modes asked
    s = 10.0 ** (-np.arange(50) / float(os.environ["SLOW"]))
n = len(s)
Q1, _ = np.linalg.qr(rng.standard_normal((M, M)))
Q2, _ = np.linalg.qr(rng.standard_normal((dim, M)))
full = np.concatenate([s, (s[-1] * 10.0 ** (-np.arange(1, M - n + 1) / float(os.environ['SLOW'])) if os.environ.get('SLOW') else 1e-15 * rng.uniform(0.3, 1, M - n))])
Xh = (Q1 * full) @ Q2.T
X = ctx.upload(Xh)
comps, sig = RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), n, center=False)
print(RB.pod_modes.last_info, "orthonormality defect", np.abs(comps @ comps.T - np.eye(n)).max())
for i in range(n):
    c = comps[i]
    ang = np.linalg.norm(c - (c @ Q2[:, i]) * Q2[:, i])
    print(f"{i:3d} sigma {s[i]:.2e} ours {sig[i]:.6e} rel {abs(sig[i] / s[i] - 1):.1e} angle {ang:.2e} noise-bound {1e-16 / s[i]:.1e}")
