"""rom_pod_ex at the C2 geometry (1024 x 65025, 50 modes) for a few relative floors (dev tool): what the exhaustive default
costs against what a reduced-basis user needs."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
fem = _ffi.Fem(ctx, 2, 2, 128)
M, n = 1024, 50
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 4))
U = ctx.alloc(M * fem.dim)
fem.solve_batch(ctx.upload(a), M, U)
host = U.download(M * fem.dim)
V = ctx.alloc(n * fem.dim)
for floor in (0.0, 1e-12, 1e-10, 1e-8, 1e-6):
    ts = []
    for rep in range(4):
        X = ctx.upload(host)
        ctx.synchronize()
        t0 = time.perf_counter()
        sig, info = ctx.pod(X, M, fem.dim, n, V, center=True, rel_floor=floor)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"rel_floor {floor:g}: {min(ts) * 1e3:.2f} ms  resolved {info['resolved_modes']}  stop {info['stop_reason']}  "
          f"sketch passes {info['sketch_passes']}  sigma_resolved_min/sigma_1 {sig[info['resolved_modes'] - 1] / sig[0]:.1e}")
