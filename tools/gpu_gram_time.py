"""Rate of rom_gram (lower 128-tiles) on snapshot-shaped blocks (dev tool).  env: MS (comma list of row counts), D."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd import _ffi

ctx = _ffi.get_context(0)
D = int(os.environ.get("D", "65025"))
for M in [int(x) for x in os.environ.get("MS", "1024,4096,8192").split(",")]:
    X = ctx.upload(np.random.default_rng(M).standard_normal((M, D)))
    G = ctx.alloc(M * M)
    ctx.gram(M, D, X, 0, D, G, 0, M)
    ctx.synchronize()
    ts = []
    for _ in range(5):
        ctx.timer_start()
        ctx.gram(M, D, X, 0, D, G, 0, M)
        ts.append(ctx.timer_stop())
    t = float(np.median(ts))
    nt = (M + 127) // 128
    executed = nt * (nt + 1) / 2 * 2.0 * 128 * 128 * D
    print(f"M={M:5d} D={D}: {t:8.3f} ms  {executed / t * 1e-9:6.1f} TFLOP/s executed (lower tiles), "
          f"{M * (M + 1.0) * D / t * 1e-9:6.1f} in M(M+1)D accounting")
    if M <= 1024:
        g = G.download(M * M).reshape(M, M)
        x = X.download(M * D).reshape(M, D)
        ref = x @ x.T
        print("      max |G - X X^T| / max|G| =", float(np.abs(g - ref).max() / np.abs(ref).max()))
