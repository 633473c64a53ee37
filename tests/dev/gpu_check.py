"""Quick GPU parity + timing probe of rom_solve_batch against the oracle (dev tool, not a test)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from oracle import rom_oracle as ro

ctx = _ffi.get_context(0)
print(ctx.device_name())
rng = np.random.default_rng(0)

def check(blocks, N, a, label):
    g = ro.Geometry(blocks, N)
    fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
    M = len(a)
    ab = ctx.upload(a.reshape(M, -1))
    U = ctx.alloc(M * g.dim)
    fem.solve_batch(ab, M, U)
    Ug = U.download(shape=(M, g.dim))
    Uo = ro.generate_solutions(g, a, "lsqsparse")
    e = ro.H10norm(g, Ug - Uo) / ro.H10norm(g, Uo)
    print(f"{label} blocks={blocks} N={N} M={M} nG={fem.n_interface} tiles={fem.n_tiles} max rel H10 err {e.max():.3e}")
    return e.max()

check((2, 2), 10, np.array([[[1, 1], [1, 1]], [[1, 1], [1, 100]], [[1, 2], [3, 4]]], float), "G1")
check((1, 1), 8, np.array([[[2.0]], [[5.0]]]), "1x1")
check((2, 2), 16, 10.0 ** rng.uniform(0, 3, size=(5, 2, 2)), "C1-like")
check((2, 3), 5, 10.0 ** rng.uniform(0, 2, size=(3, 2, 3)), "rect")
check((3, 2), 4, 10.0 ** rng.uniform(0, 2, size=(3, 3, 2)), "rect")
check((1, 3), 7, 10.0 ** rng.uniform(0, 2, size=(3, 1, 3)), "strip")
check((3, 3), 11, 10.0 ** rng.uniform(0, 8, size=(4, 3, 3)), "3x3 hc")
check((4, 4), 8, 10.0 ** rng.uniform(0, 8, size=(4, 4, 4)), "4x4 hc")
check((2, 2), 70, 10.0 ** rng.uniform(0, 2, size=(3, 2, 2)), "2 tiles/edge")
check((2, 2), 128, 10.0 ** rng.uniform(0, 2, size=(2, 2, 2)), "C2 size")
check((3, 3), 40, 10.0 ** rng.uniform(0, 4, size=(2, 3, 3)), "3x3 N40")

# timing at C2
blocks, N, M = (2, 2), 128, 1024
g = ro.Geometry(blocks, N)
t0 = time.time()
fem = _ffi.Fem(ctx, 2, 2, N)
ctx.synchronize()
print("fem_create s", time.time() - t0)
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 2, 2))
ab = ctx.upload(a.reshape(M, -1))
U = ctx.alloc(M * g.dim)
fem.solve_batch(ab, M, U)
ctx.synchronize()
ctx.profile(True)
for rep in range(3):
    ctx.timer_start()
    fem.solve_batch(ab, M, U)
    ms = ctx.timer_stop()
    print(f"solve_batch M={M}: {ms:.3f} ms -> {M / ms * 1e3:.0f} solves/s")
rep = ctx.profile_report()
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"  {k:20s} {v['total_ms']/3:9.3f} ms/step  launches/step {v['launches']/3:6.1f}  "
          f"{v['flops']/v['total_ms']*1e-9 if v['total_ms'] else 0:8.2f} TFLOP/s  {v['bytes']/v['total_ms']*1e-6 if v['total_ms'] else 0:8.1f} GB/s")
print(fem.solve_work())
idx = [0, 511, 1023]
Uo = ro.generate_solutions(g, a[idx], "lsqsparse")
Ug = np.stack([U.download(g.dim, offset=i * g.dim) for i in idx])
print("C2 parity (3 rows):", (ro.H10norm(g, Ug - Uo) / ro.H10norm(g, Uo)))
