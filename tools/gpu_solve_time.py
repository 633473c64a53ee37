"""Timing probe of the reduced solve alone (dev tool): rom_solve_reduced_async on coefficients that are already in HBM,
per-kernel times from the library's own profile records.  env: NB, N, M, REPS, INNER."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd import _ffi

ctx = _ffi.get_context(0)
NB, N, M = int(os.environ.get("NB", "3")), int(os.environ.get("N", "171")), int(os.environ.get("M", "1024"))
reps, inner = int(os.environ.get("REPS", "5")), int(os.environ.get("INNER", "10"))
fem = _ffi.Fem(ctx, NB, NB, N)
a = 10.0 ** np.random.default_rng(20240807).uniform(0, float(os.environ.get("DEC", "2")), size=(M, NB * NB))
ab = ctx.upload(a)
Y = ctx.alloc(M * fem.reduced_stride)
fem.solve_reduced(ab, M, Y)
ctx.synchronize()
ts = []
for rep in range(reps):
    ctx.timer_start()
    for _ in range(inner):
        fem.solve_reduced(ab, M, Y)
    ts.append(ctx.timer_stop() / inner)
print(f"reduced solve {NB}x{NB} N={N} M={M}: min {min(ts):.4f} ms  median {np.median(ts):.4f} ms")
ctx.profile(True)
ctx.profile_reset()
for _ in range(inner):
    fem.solve_reduced(ab, M, Y)
ctx.synchronize()
for name, rec in sorted(ctx.profile_report().items()):
    print(f"   {name:16s} {rec['launches'] / inner:5.1f} launches  {rec['total_ms'] / inner:8.4f} ms per solve")
ctx.profile(False)
