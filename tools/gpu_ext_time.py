"""Timing probe of the extension stage alone (dev tool): rom_expand_batch_async on interface vectors that are
already in HBM, variants selected by environment switches and run INTERLEAVED in one process (rule 24 of the HIP
guide: A/B deltas come from interleaved rounds).  env: NB, N, M, REPS; VARIANTS = ';'-separated 'name:ENV=1,ENV2=0'."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd import _ffi

if os.environ.get('ROMHC_LIB'):
    _ffi.load_library(os.path.abspath(os.environ['ROMHC_LIB']))  # (dev: a variant build)
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
fems = {}
for name, env in variants:  # (the switches are read once per FE space: one per variant)
    for k, v in env.items():
        os.environ[k] = v
    fems[name] = _ffi.Fem(ctx, NB, NB, N)
    for k in env:
        del os.environ[k]
for rep in range(reps):
    for name, env in variants:
        fem = fems[name]
        fem.expand(ab, M, Y, U)  # warm
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(inner):
            fem.expand(ab, M, Y, U)
        times[name].append(ctx.timer_stop() / inner)
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

