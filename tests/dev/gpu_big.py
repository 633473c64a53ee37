"""Full-size functional probe of configs C4 / C5 (dev tool): setup time, sweep throughput, residual parity."""
import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from oracle import rom_oracle as ro

ctx = _ffi.get_context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
if which == "c4":
    blocks, N, M = (3, 3), 171, 256
    rng = np.random.default_rng(20240807)
    a = np.ones((M, 3, 3))
    for j in range(9):
        a[1 + j].flat[j] = 1e8
    a[10] = 1e8
    a[11:] = 10.0 ** rng.uniform(0, 8, size=(M - 11, 3, 3))
else:
    blocks, N, M = (4, 4), 256, int(os.environ.get("M", "64"))
    a = 10.0 ** np.random.default_rng(20240807).uniform(0, 3, size=(M, 4, 4))
t0 = time.time()
fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
ctx.synchronize()
print(f"{which}: blocks={blocks} N={N} dim={fem.dim} nG={fem.n_interface} tiles={fem.n_tiles} setup {time.time()-t0:.2f}s", flush=True)
print(fem.solve_work(), flush=True)
ab = ctx.upload(a.reshape(M, -1))
U = ctx.alloc(M * fem.dim)
fem.solve_batch(ab, M, U)
ctx.synchronize()
ctx.profile(True)
ctx.timer_start()
fem.solve_batch(ab, M, U)
ms = ctx.timer_stop()
print(f"solve_batch M={M}: {ms:.1f} ms -> {M/ms*1e3:.1f} solves/s", flush=True)
for k, v in sorted(ctx.profile_report().items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"  {k:20s} {v['total_ms']:9.3f} ms  launches {v['launches']:5d}  {v['flops']/v['total_ms']*1e-9 if v['total_ms'] else 0:8.2f} TFLOP/s")
ctx.profile(False)
# residual check through the independent stencil kernel, plus one oracle row
Y = ctx.alloc(fem.dim)
B = fem.load_vector()
for m in (0, 1, 5, 10, M - 1):
    row = _ffi.Buffer(ctx, fem.dim).copy_from(U, fem.dim, 0, m * fem.dim)
    fem.stencil_apply(row, 1, Y, a_one=a[m].ravel())
    r = Y.download(fem.dim) - B
    u = row.download(fem.dim)
    print(f"  row {m}: max|A u - b| / (max|a| max|u|) = {np.abs(r).max() / (a[m].max() * np.abs(u).max()):.2e}", flush=True)
g = ro.Geometry(blocks, N)
m = M - 1
t0 = time.time()
uo = ro.solve_one(g, a[m], B, "lsqsparse")
ug = U.download(fem.dim, offset=m * fem.dim)
print(f"  oracle row {m} ({time.time()-t0:.1f}s CPU): rel H10 err {ro.H10norm(g, (ug-uo)[None])[0]/ro.H10norm(g, uo[None])[0]:.2e}")
