"""k_extend_lines (whole-line stores) against k_extend128: bit identity of the snapshot rows over geometries / batch sizes, and
per-kernel times (dev tool).  usage: python tools/dev/gpu_extend_lines.py [check] [time] [sweep]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi  # noqa: E402

ctx = _ffi.get_context(0)
what = set(sys.argv[1:]) or {"check", "time"}


def fem_with(env, blocks, N):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _ffi.Fem(ctx, blocks[0], blocks[1], N)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def rows(fem, ab, M, row0=0, extra=0):
    U = ctx.alloc((M + row0 + extra) * fem.dim)
    U.fill(float("nan"))
    fem.solve_batch(ab, M, U, row0)
    return U.download(shape=(M + row0 + extra, fem.dim))


if "check" in what:
    bad = 0
    for blocks, N, M in (((2, 2), 128, 1024), ((2, 2), 128, 130), ((2, 2), 128, 64), ((2, 2), 128, 65), ((2, 2), 128, 257),
                         ((1, 2), 128, 100), ((2, 1), 128, 70), ((2, 3), 128, 200), ((3, 2), 128, 96), ((4, 4), 256, 72),
                         ((2, 2), 256, 130)):
        a = 10.0 ** np.random.default_rng(N + M).uniform(0, 3, size=(M, blocks[0] * blocks[1]))
        ab = ctx.upload(a)
        ref = rows(fem_with({"ROMHC_EXT_LINES": "0"}, blocks, N), ab, M)
        for env in ({}, {"ROMHC_LINES_ROWS": "3"}, {"ROMHC_LINES_ORDER": "1", "ROMHC_LINES_ROWS": "7"}):
            got = rows(fem_with(env, blocks, N), ab, M)
            same = np.array_equal(got, ref)
            nbad = int((got != ref).sum()) if not same else 0
            nan = int(np.isnan(got).sum())
            print(f"blocks={blocks} N={N} M={M} {env}: identical={same} differing={nbad} nan={nan}", flush=True)
            if not same:
                bad += 1
                idx = np.argwhere(got != ref)
                print("   first differing (system, dof):", idx[:8].tolist())
                r, c = idx[0]
                print("   values", got[r, c], ref[r, c], " dof -> mesh row/col", c // (blocks[1] * N - 1), c % (blocks[1] * N - 1))
    print("CHECK", "FAILED" if bad else "ok", flush=True)

if "time" in what or "sweep" in what or "probe" in what:
    blocks, N, M = (2, 2), 128, 1024
    a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 4))
    ab = ctx.upload(a)
    variants = [("k_extend128", {"ROMHC_EXT_LINES": "0"}), ("lines default", {})]
    if "probe" in what:
        for d in [int(x) for x in os.environ.get('PROBES', '1,2,4,6,8,14,15').split(',')]:
            variants.append((f"lines dbg={d}", {"ROMHC_LINES_DBG": str(d)}))
    if "sweep" in what:
        for R in (16, 8, 4):
            for order in (0, 1):
                variants.append((f"lines R={R} order={order}", {"ROMHC_LINES_ROWS": str(R), "ROMHC_LINES_ORDER": str(order)}))
    for rep in range(2):
        for name, env in variants:
            fem = fem_with(env, blocks, N)
            U = ctx.alloc(M * fem.dim)
            for _ in range(20):
                fem.solve_batch(ab, M, U, wait=False)
            ctx.synchronize()
            ctx.profile_reset()
            ctx.profile(True)
            for _ in range(50):
                fem.solve_batch(ab, M, U, wait=False)
            ctx.synchronize()
            ctx.profile(False)
            prof = ctx.profile_report()
            t0 = time.perf_counter()
            for _ in range(200):
                fem.solve_batch(ab, M, U, wait=False)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / 200
            ks = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in prof.items()}
            print(f"{name:28s} step {dt * 1e3:.4f} ms   kernels (ms per launch): {ks}", flush=True)
