"""Timing probe of the extension stage alone (dev tool): rom_expand_batch_async on interface vectors that are
already in HBM, variants selected by environment switches and run INTERLEAVED in one process (rule 24 of the HIP
guide: A/B deltas come from interleaved rounds).  env: NB, N, M, REPS; VARIANTS = ';'-separated 'name:ENV=1,ENV2=0'."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd import _ffi

ctx = _ffi.get_context(0)
NB, N, M = int(os.environ.get("NB", "2")), int(os.environ.get("N", "128")), int(os.environ.get("M", "1024"))
reps, inner = int(os.environ.get("REPS", "7")), int(os.environ.get("INNER", "20"))
fem = _ffi.Fem(ctx, NB, NB, N)
a = 10.0 ** np.random.default_rng(20240807).uniform(0, float(os.environ.get("DEC", "2")), size=(M, NB * NB))
ab, U = ctx.upload(a), ctx.alloc(M * max(fem.dim, fem.nr * 256 if os.environ.get('PADTEST') else 0))
Y = ctx.alloc(M * fem.reduced_stride)
fem.solve_reduced(ab, M, Y)
ctx.solve_status()
variants = [("default", {})]
for spec in filter(None, os.environ.get("VARIANTS", "").split(";")):
    name, _, envs = spec.partition(":")
    variants.append((name, dict(kv.split("=") for kv in envs.split(",") if kv)))
ref = None
times = {n: [] for n, _ in variants}
for rep in range(reps):
    for name, env in variants:
        for k, v in env.items():
            os.environ[k] = v
        fem.expand(ab, M, Y, U)  # warm
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(inner):
            fem.expand(ab, M, Y, U)
        times[name].append(ctx.timer_stop() / inner)
        for k in env:
            del os.environ[k]
        if rep == 0:
            out = U.download(8 * fem.dim)
            if ref is None:
                ref = out
            else:
                print(f"   {name}: rows identical to default: {np.array_equal(out, ref)}")
fl = fem.solve_work()["flops_own"] * M
for name, _ in variants:
    t = np.array(times[name])
    print(f"{name:24s} min {t.min():.4f} ms  median {np.median(t):.4f} ms  -> {fl / np.median(t) * 1e-9:.2f} TFLOP/s algorithmic")

if os.environ.get("CLOCKS"):
    import ctypes
    lib = _ffi.load_library()
    for name, env in variants:
        for k, v in env.items():
            os.environ[k] = v
        cz = (ctypes.c_int * 2)()
        if hasattr(lib, "rom_debug_xs_census"):
            lib.rom_debug_xs_census(cz, 1)
        for _ in range(30):
            fem.expand(ab, M, Y, U)
        ctx.synchronize()
        if hasattr(lib, "rom_debug_xs_census"):
            lib.rom_debug_xs_census(cz, 1)
            print(f"{name:24s} workgroups inside the kernel at once: max {cz[1]} (left over {cz[0]})")
        for k in env:
            del os.environ[k]
        buf = (ctypes.c_ulonglong * (1024 * 4))()
        if hasattr(lib, "rom_debug_xs_clock") and lib.rom_debug_xs_clock(buf, 1024 * 4) == 0:
            c = np.array(buf[:]).reshape(1024, 4).astype(np.float64)
            ok = (c[:, 3] > c[:, 1]) & (c[:, 2] > c[:, 0])
            ghz = (c[ok, 2] - c[ok, 0]) / (c[ok, 3] - c[ok, 1]) * 0.1
            us = (c[ok, 3] - c[ok, 1]) / 100.0
            print(f"{name:24s} in-kernel clock: median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}); "
                  f"workgroup lifetime median {np.median(us):.1f} us, max {us.max():.1f} us, min {us.min():.1f} us")
