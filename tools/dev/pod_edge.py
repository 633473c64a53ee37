import os, sys, logging
import numpy as np
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from romhighcontrast_amd import _ffi
logging.disable(logging.WARNING)
ctx = _ffi.get_context()
rng = np.random.default_rng(3)
def run(name, Xh, n, center=True):
    M, dim = Xh.shape
    Xc = Xh - Xh.mean(axis=0) if center else Xh
    _, sv, Vt = np.linalg.svd(Xc, full_matrices=False)
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), M, dim), n, center=center)
    info = RB.pod_modes.last_info
    k = int((sv[:n] > 1e-9 * sv[0]).sum())
    rel = np.abs(sig[:k] / sv[:k] - 1).max() if k else 0.0
    P1, P2 = comps[:k].T @ comps[:k], Vt[:k].T @ Vt[:k]
    print(f"{name}: M {M} dim {dim} n {n}: resolved {info['resolved_modes']} gram {info['gram_passes']} passes {info['sketch_passes']} stop {info['stop_reason']}; "
          f"sv rel err (>{1e-9:g} s1, {k} modes) {rel:.1e}; projector diff {np.abs(P1 - P2).max():.1e}; orth {np.abs(comps @ comps.T - np.eye(n)).max():.1e}; sv ratio last {sv[min(n, len(sv)) - 1] / sv[0]:.1e}")
# huge mean
s = 10.0 ** -np.arange(0, 10, 0.5)
Q1, _ = np.linalg.qr(rng.standard_normal((300, 300)))
Q2, _ = np.linalg.qr(rng.standard_normal((4000, 300)))
base = (Q1[:, :len(s)] * s) @ Q2[:, :len(s)].T
for mean_scale in (1.0, 1e3, 1e6):
    run(f"mean x{mean_scale:g}", base + mean_scale * rng.standard_normal(4000)[None, :], 12)
# rank deficient, duplicates
run("rank 5, n 10", (Q1[:, :5] * [1, .5, .1, .01, .001]) @ Q2[:, :5].T, 10, center=False)
run("all rows equal", np.tile(rng.standard_normal(2000), (40, 1)), 5)
run("M=2", rng.standard_normal((2, 500)), 2)
run("M=3 n=3 centred", rng.standard_normal((3, 500)), 3)
run("dim < M", rng.standard_normal((400, 60)) @ np.diag(10.0 ** -np.linspace(0, 6, 60)), 40, center=False)
run("n = M", rng.standard_normal((24, 3000)), 24, center=False)
Xh = (Q1[:, :40] * np.repeat([1.0, 1e-3, 1e-6, 1e-9], 10)) @ Q2[:, :40].T
comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), 300, 4000), 40, center=False)
print("clustered:", RB.pod_modes.last_info["sketch_passes"], "passes; sigma rel err", np.abs(sig / np.repeat([1.0, 1e-3, 1e-6, 1e-9], 10) - 1).max())
for c in range(4):
    P1 = comps[10 * c:10 * c + 10].T @ comps[10 * c:10 * c + 10]
    Vc = Q2[:, 10 * c:10 * c + 10]
    print("  cluster", c, "projector diff", np.abs(P1 - Vc @ Vc.T).max(), "noise bound", 1.1e-16 / [1.0, 1e-3, 1e-6, 1e-9][c])
