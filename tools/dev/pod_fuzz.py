"""rom_pod against numpy.linalg.svd on random shapes / spectra / requests (dev probe).  env: CASES (60), SEED (0)."""
import os, sys, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from romhighcontrast_amd import _ffi
logging.disable(logging.WARNING)
ctx = _ffi.get_context()
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
bad = 0
for case in range(int(os.environ.get("CASES", "60"))):
    M = int(rng.integers(2, int(os.environ.get("MMAX", "400"))))
    dim = int(rng.integers(2, int(os.environ.get("DMAX", "5000"))))
    r = min(M, dim)
    n = int(rng.integers(1, r + 1)) if rng.random() < 0.3 else int(rng.integers(1, min(r, 60) + 1))
    center = bool(rng.integers(0, 2))
    kind = rng.integers(0, 4)
    if kind == 0:      # geometric decay, random rate
        s = 10.0 ** (-np.arange(r) * rng.uniform(0.02, 1.5))
    elif kind == 1:    # plateau + cliff
        k = int(rng.integers(1, r + 1))
        s = np.concatenate([np.ones(k), 1e-9 * np.ones(r - k)])
    elif kind == 2:    # random gaussian block (flat-ish spectrum)
        s = None
    else:              # low rank
        k = int(rng.integers(1, min(r, 12) + 1))
        s = np.concatenate([10.0 ** -rng.uniform(0, 6, k), np.zeros(r - k)])
    if s is None:
        Xh = rng.standard_normal((M, dim))
    else:
        Q1, _ = np.linalg.qr(rng.standard_normal((M, r)))
        Q2, _ = np.linalg.qr(rng.standard_normal((dim, r)))
        Xh = (Q1 * s) @ Q2.T
    if center:
        Xh = Xh + rng.uniform(0, 3) * rng.standard_normal(dim)[None, :]
    Xc = Xh - Xh.mean(axis=0) if center else Xh
    if os.environ.get("ONLY") and case != int(os.environ["ONLY"]):
        continue
    sv = np.linalg.svd(Xc, compute_uv=False)
    import time
    t0 = time.perf_counter()
    comps, sig = RB.pod_modes(ctx, SM.DeviceArray(ctx.upload(Xh), M, dim), n, center=center)
    dt = time.perf_counter() - t0
    if os.environ.get("ONLY"): print("sv", sv[:16], "\nsig", sig[:16], sig[-3:])
    if dt > 0.2: print(f"case {case}: M {M} dim {dim} n {n} kind {kind}: {dt:.2f} s", RB.pod_modes.last_info)
    info = RB.pod_modes.last_info
    floor = 50 * 1.1e-16 * np.linalg.norm(Xh, 2)
    m = min(n, len(sv))
    err = np.abs(sig[:m] - sv[:m])
    tol = 1e-7 * sv[:m] + floor
    # completed modes (sigma = 0) are allowed where LAPACK's value is below the POD's floor 1e-13 sigma_1 (+ noise)
    okv = np.all((err <= tol) | ((sig[:m] == 0) & (sv[:m] <= 2e-13 * sv[0] + floor)))
    orth = np.abs(comps @ comps.T - np.eye(n)).max()
    ok = okv and orth < 1e-12
    if not ok:
        bad += 1
        w = int(np.argmax(err / tol))
        print(f"CASE {case} FAIL: M {M} dim {dim} n {n} center {center} kind {kind}: worst at {w}: sv {sv[w]:.3e} ours {sig[w]:.3e} (sv0 {sv[0]:.3e}); orth {orth:.1e}; info {info}")
print("cases failed:", bad)
