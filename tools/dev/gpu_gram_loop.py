"""Keeps rom_gram busy for a few seconds (dev tool, for tools/dev/gpu_power_probe.sh). env: M, D, SECONDS."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
M, D, secs = int(os.environ.get("M", "4096")), int(os.environ.get("D", "65025")), float(os.environ.get("SECONDS", "5"))
X = ctx.upload(np.random.default_rng(M).standard_normal((M, D)))
G = ctx.alloc(M * M)
ctx.gram(M, D, X, 0, D, G, 0, M); ctx.synchronize()
t0 = time.time(); ts = []
while time.time() - t0 < secs:
    ctx.timer_start()
    for _ in range(5): ctx.gram(M, D, X, 0, D, G, 0, M)
    ts.append(ctx.timer_stop() / 5)
t = float(np.median(ts))
print(f"rom_gram M={M} D={D}: {t:.3f} ms, {M * (M + 1.0) * D / t * 1e-9:.1f} TFLOP/s (M(M+1)D)")
