import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context()
rng = np.random.default_rng(1)
for n in (97, 200, 511, 1000, 1024):
    B = rng.standard_normal((n, n + 50))
    A = B @ B.T
    t0 = time.perf_counter()
    lam, T = ctx.small_eig(A, mode=0, gram_like=True)
    dt = time.perf_counter() - t0
    w = np.linalg.eigvalsh(A)[::-1]
    print(n, f"{dt:.3f} s", "rel eig err", np.abs(lam / w - 1).max(), "orth", np.abs(T @ T.T - np.eye(n)).max(), "resid", np.abs(T @ A @ T.T - np.diag(lam)).max() / w[0])
    # graded
    d = 10.0 ** -np.linspace(0, 12, n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Q * d) @ Q.T; A = 0.5 * (A + A.T)
    lam, T = ctx.small_eig(A, mode=0, gram_like=False)
    w = np.linalg.eigvalsh(A)[::-1]
    print("   graded: abs eig err / lam1", np.abs(lam - w).max(), "orth", np.abs(T @ T.T - np.eye(n)).max())
