"""Timing probe for rom_solve_batch (dev tool): env NB (blocks per side, default 2), N (cells per block), M."""
import sys, os
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
NB = int(os.environ.get("NB", "2"))
N, M = int(os.environ.get("N", "128")), int(os.environ.get("M", "1024"))
fem = _ffi.Fem(ctx, NB, NB, N)
a = 10.0 ** np.random.default_rng(20240807).uniform(0, float(os.environ.get("DEC", "2")), size=(M, NB, NB))
ab = ctx.upload(a.reshape(M, -1))
U = ctx.alloc(M * fem.dim)
for _ in range(3):
    fem.solve_batch(ab, M, U)
ctx.synchronize()
best = 1e9
for rep in range(10):
    ctx.timer_start()
    fem.solve_batch(ab, M, U)
    best = min(best, ctx.timer_stop())
print(f"streams={os.environ.get('ROMHC_STREAMS','default')} M={M}: best {best:.3f} ms -> {M / best * 1e3:.0f} solves/s")
if os.environ.get("PROFILE"):
    ctx.profile(True)
    for rep in range(3):
        fem.solve_batch(ab, M, U)
    rep = ctx.profile_report()
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"  {k:20s} {v['total_ms']/3:9.3f} ms/step  launches/step {v['launches']/3:6.1f}  "
              f"{v['flops']/v['total_ms']*1e-9 if v['total_ms'] else 0:8.2f} TFLOP/s")
