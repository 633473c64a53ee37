"""Where the GPU rows of high-contrast parameters differ from the extended-precision truth (dev probe, round 4):
 (1) fixture g8's INFINIT_A rows ((2,2)/N=6): error by region (blocks / interface) against tests/golden/referee_g8_inf.npz;
 (2) the floating-block row of fixture g4 ((3,3), centre block at 1e10) at N = 11, 22, 44, 88: GPU, SuperLU (lsqsparse) and
     LAPACK posv (lsq, small N only) against the truth -- how each solver's error moves with N."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rom_oracle as ro  # noqa: E402
from referee import LD, h10_ld, referee  # noqa: E402
from romhighcontrast_amd import _ffi  # noqa: E402

ctx = _ffi.get_context(0)


def gpu_rows(blocks, N, a):
    fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
    M = len(a)
    U = ctx.alloc(M * fem.dim)
    fem.solve_batch(ctx.upload(a.reshape(M, -1)), M, U)
    return U.download(shape=(M, fem.dim))


if "g8" in sys.argv[1:] or len(sys.argv) == 1:
    z = np.load(os.path.join(ROOT, "tests/golden/g8_experiment.npz"), allow_pickle=True)
    r8 = np.load(os.path.join(ROOT, "tests/golden/referee_g8_inf.npz"))
    a = z["a"]
    U = gpu_rows((2, 2), 6, a)
    g = ro.Geometry((2, 2), 6)
    for k, row in enumerate(r8["rows"]):
        t = r8["truth"][k].reshape(g.nr, g.nc)
        u = U[row].reshape(g.nr, g.nc)
        d = np.abs(u - t)
        e = float(h10_ld(g, (U[row] - r8["truth"][k]).astype(LD)) / h10_ld(g, r8["truth"][k].astype(LD)))
        reg = {"b00": d[:5, :5].max(), "b01": d[:5, 6:].max(), "b10": d[6:, :5].max(), "b11": d[6:, 6:].max(),
               "h-iface": d[5, :].max(), "v-iface": d[:, 5].max()}
        mag = {"b00": np.abs(t[:5, :5]).max(), "b01": np.abs(t[:5, 6:]).max(), "b10": np.abs(t[6:, :5]).max(), "b11": np.abs(t[6:, 6:]).max(),
               "h-iface": np.abs(t[5, :]).max(), "v-iface": np.abs(t[:, 5]).max()}
        print(f"g8 row {row} a={a[row].ravel().tolist()} relH10 {e:.2e} (reference {r8['err_ref_vs_truth'][k]:.1e})")
        print("    max |err| by region:", {k2: f"{v:.1e}" for k2, v in reg.items()})
        print("    max |u|   by region:", {k2: f"{v:.1e}" for k2, v in mag.items()}, flush=True)

if "b33" in sys.argv[1:] or len(sys.argv) == 1:
    z4 = np.load(os.path.join(ROOT, "tests/golden/referee_g4_floating.npz"))
    a = z4["b33_a"]
    print("b33 row", a.tolist())
    for N in (11, 22, 44, 88):
        g, truth, x0, err_slu, hist = referee((3, 3), N, a, verbose=False)
        tl = truth.astype(LD)
        ug = gpu_rows((3, 3), N, a[None])[0]
        e_gpu = float(h10_ld(g, ug.astype(LD) - tl) / h10_ld(g, tl))
        e_lsq = float("nan")
        if N <= 44:
            ul = ro.generate_solutions(g, a[None], "lsq")[0]
            e_lsq = float(h10_ld(g, ul.astype(LD) - tl) / h10_ld(g, tl))
        print(f"(3,3) N={N:3d} dim {g.dim:6d}: GPU {e_gpu:.3e}   SuperLU {err_slu:.3e}   LAPACK posv {e_lsq:.3e}   (contrast x eps = {1e10 * 2.2e-16:.1e})", flush=True)
