"""Reader of the in-kernel cycle stamps of k_solve1 (dev tool; needs a library built with
`make EXTRA=-DROMHC_SOLVE1_STAMPS`, whose interface vectors carry the stamps instead of the first unknowns).
usage: python tools/dev/gpu_solve1_stamps.py path/to/libromhc_stamps.so"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd import _ffi

if len(sys.argv) > 1:
    _ffi.load_library(os.path.abspath(sys.argv[1]))
ctx = _ffi.get_context(0)
M = int(os.environ.get("M", "1024"))
fem = _ffi.Fem(ctx, 2, 2, 128)
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 4))
ab = ctx.upload(a)
Y = ctx.alloc(M * fem.reduced_stride)
names = ["prologue", "assembly", "panel 0 + barrier", "rhs", "cholesky", "back subst", "-", "-", "-"]
for rep in range(3):
    fem.solve_reduced(ab, M, Y)
    ctx.solve_status()
    ctx.timer_start()
    for _ in range(20):
        fem.solve_reduced(ab, M, Y)
    ms = ctx.timer_stop() / 20
    y = Y.download(M * fem.reduced_stride).reshape(M, -1)
    st = y[:, :10]
    d = np.diff(st, axis=1)
    print(f"rep {rep}: kernel {ms * 1e3:.1f} us; wave total {st[:, 6].mean():.0f} cycles (min {st[:, 6].min():.0f} max {st[:, 6].max():.0f});"
          f" Cholesky (-DROMHC_SOLVE1_PANEL_STAMPS): factor wave waits at the panel barriers {y[:, 10].mean():.0f}; update wave: loop {y[:, 13].mean():.0f}, of it waiting {y[:, 12].mean():.0f}")
    print("   " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(axis=0))))
    if M > 1024:
        for lo in range(0, M, 1024):
            dd = np.diff(st[lo:lo + 1024], axis=1).mean(axis=0)
            print(f"   systems {lo}..: total {st[lo:lo + 1024, 9].mean():.0f}  " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, dd)))
