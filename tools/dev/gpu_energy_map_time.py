import sys, time
sys.path.insert(0, ".")
import numpy as np
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
for blocks, N in (((3, 3), 171), ((2, 2), 128)):
    for parts in (1, 3, 7):
        fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
        ctx.synchronize()
        ctx.profile_reset(); ctx.profile(True)
        t0 = time.perf_counter()
        ranks = fem.energy_map(parts)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ctx.profile(False)
        prof = ctx.profile_report()
        top = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:6]
        print(blocks, N, "parts", parts, "ranks", ranks, f"{dt*1e3:.1f} ms", {k: (round(v["total_ms"], 2), v["launches"]) for k, v in top}, flush=True)
