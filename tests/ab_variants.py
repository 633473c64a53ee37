"""A/B checks that need the forcing switches of libromhc_ab.so (csrc/Makefile target `ab`: rom_fem_setup.hip compiled with
-DROMHC_AB): the product chooses extension tilings, workgroup orders, systems per workgroup and the tile-assembly form BY
GEOMETRY; this script forces each choice on one geometry and asserts that all of them give the same snapshot rows, bit for
bit.  Run by tests/test_gpu_parity.py::test_ab_build_variants_agree in a subprocess (a process loads one library):

    python tests/ab_variants.py            exit code 0 and a last line "OK" on success
TEST INFRASTRUCTURE (imports the oracle for the small geometries)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from romhighcontrast_amd import _ffi  # noqa: E402

_ffi.load_library(os.path.join(ROOT, "romhighcontrast_amd", "csrc", "libromhc_ab.so"))
from oracle import rom_oracle as ro  # noqa: E402


def relh10(g, U, Uref):
    return ro.H10norm(g, U - Uref) / ro.H10norm(g, Uref)


def sweep(ctx, blocks, N, ab, M, env):
    for k, v in env.items():
        os.environ[k] = v
    try:
        fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)  # (the switches are read once per FE space)
        U = ctx.alloc(M * fem.dim)
        U.fill(float("nan"))
        fem.solve_batch(ab, M, U)
        return U.download(shape=(M, fem.dim))
    finally:
        for k in env:
            del os.environ[k]


def tilings(ctx):
    """The extension into the blocks has three tilings (128 vertices of one mesh row, 128 consecutive vertices of the block,
    64 vertices of one mesh row) and three workgroup orders: same products in the same order, so the rows must be identical.
    (5 x 4 blocks: more than the 16 block descriptors one launch of the 128-tile kernel carries; N = 65 / 66: mesh rows of
    64 / 65 vertices; M = 130, 257: a last system group of two systems / one system.)"""
    for blocks, N, M in (((2, 2), 128, 130), ((3, 3), 24, 140), ((2, 3), 40, 200), ((2, 2), 90, 128), ((1, 2), 9, 129),
                         ((3, 2), 171, 128), ((5, 4), 33, 128), ((2, 2), 65, 257), ((2, 2), 66, 128), ((1, 2), 128, 384),
                         ((2, 2), 128, 1024)):
        a = 10.0 ** np.random.default_rng(N).uniform(0, 3, size=(M, blocks[0] * blocks[1]))
        ab = ctx.upload(a)
        out = {name: sweep(ctx, blocks, N, ab, M, env) for name, env in (
            ("row", {"ROMHC_EXT_FLAT": "0"}), ("flat", {"ROMHC_EXT_FLAT": "1"}), ("t64", {"ROMHC_NO_EXT128": "1"}),
            ("sysfast", {"ROMHC_X128_SYS_FAST": "1"}), ("xcd", {"ROMHC_X128_SYS_FAST": "2"}), ("nofold", {"ROMHC_NO_FOLD_EXPAND": "1"}),
            ("default", {}))}
        for name in ("flat", "t64", "sysfast", "xcd", "nofold", "default"):
            assert np.array_equal(out[name], out["row"]), (blocks, N, M, name)
        if N <= 40:
            g = ro.Geometry(blocks, N)
            assert relh10(g, out["default"][:3], ro.generate_solutions(g, a[:3].reshape((3,) + blocks))).max() < 1e-11
        print(f"tilings {blocks} N={N} M={M}: identical", flush=True)


def forced_forms(ctx):
    """One system per workgroup in the diagonal update, tiles assembled in registers, the general extension kernel for blocks
    whose sides are all compressed: forms the product uses on OTHER geometries (more than 2048 systems, tiles of more than
    128 terms, blocks with a side in sine modes; ROMHC_COEF_GLOBAL: k_coef's dot products in HBM, the form of geometries whose
    closed-form entries outgrow 156 KB of LDS), forced here."""
    for env in ("ROMHC_NO_EXT_LR", "ROMHC_NO_TILE_PAIRS", "ROMHC_NO_TILE_STREAM", "ROMHC_COEF_GLOBAL"):
        for blocks, N, M in (((2, 2), 128, 130), ((3, 3), 24, 40), ((2, 3), 40, 20), ((3, 3), 64, 70)):
            a = 10.0 ** np.random.default_rng(N).uniform(0, 3, size=(M, blocks[0] * blocks[1]))
            ab = ctx.upload(a)
            ref, alt = sweep(ctx, blocks, N, ab, M, {}), sweep(ctx, blocks, N, ab, M, {env: "1"})
            g = ro.Geometry(blocks, N)
            assert relh10(g, alt, ref).max() < 1e-11, (env, blocks, N)
            if env != "ROMHC_NO_EXT_LR":  # (same sums in the same order)
                assert np.array_equal(alt, ref), (env, blocks, N)
            if N <= 40:
                assert relh10(g, alt[:4], ro.generate_solutions(g, a[:4].reshape((4,) + blocks))).max() < 1e-11
        print(f"{env}: agrees", flush=True)


if __name__ == "__main__":
    ctx = _ffi.get_context()
    tilings(ctx)
    forced_forms(ctx)
    print("OK")
