"""Config C4 end to end on the device (dev tool): 1024-snapshot sweep at 512^2 (3x3 blocks, contrast 1e8),
H10 norms, greedy reduced basis to dim n for both greedy modes, error decay."""
import sys, time, os
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
from romhighcontrast_amd.lib.SolutionsManagers import SolutionsManagerFEM
from romhighcontrast_amd.lib import ReducedBasis as RB
ctx = _ffi.get_context(0)
blocks, N = (3, 3), int(os.environ.get("N", "171"))
M, n = int(os.environ.get("M", "1024")), int(os.environ.get("NB", "50"))
rng = np.random.default_rng(20240807)
a = np.ones((M, 3, 3))
for j in range(9):
    a[1 + j].flat[j] = 1e8
a[10] = 1e8
a[11:] = 10.0 ** rng.uniform(0, 8, size=(M - 11, 3, 3))
t0 = time.perf_counter(); sm = SolutionsManagerFEM(blocks, N); ctx.synchronize(); print(f"setup {time.perf_counter()-t0:.2f}s dim={sm.vspace_dim}")
def T(f):
    ctx.synchronize(); t = time.perf_counter(); out = f(); ctx.synchronize(); return out, time.perf_counter() - t
U, t = T(lambda: sm.generate_solutions_device(a)); print(f"sweep M={M}: {t*1e3:.1f} ms -> {M/t:.0f} solves/s")
U, t = T(lambda: sm.generate_solutions_device(a)); print(f"sweep M={M}: {t*1e3:.1f} ms -> {M/t:.0f} solves/s (warm)")
h1, t = T(lambda: sm.H10norm(U)); h1, t = T(lambda: sm.H10norm(U))
print(f"H10norm of {M} vectors: {t*1e3:.2f} ms -> {8.0*M*sm.vspace_dim/t*1e-9:.0f} GB/s algorithmic (8 B/entry read once)")
for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
    rb = RB.ReducedBasisGreedy(mode)
    if os.environ.get("PROFILE"):
        ctx.profile(True)
    _, t = T(lambda: rb.build(n, sm, U, a, h1))
    if os.environ.get("PROFILE"):
        rep = ctx.profile_report()
        ctx.profile(False)
        tot = sum(v["total_ms"] for v in rep.values())
        print(f"   kernels of the build: {tot:.1f} ms in total")
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])[:9]:
            print(f"     {k:18s} {v['total_ms']:8.2f} ms  launches {v['launches']:6d}")
    e = np.array(rb.max_errors)
    print(f"greedy {mode}: n={n} in {t:.2f} s; max rel H10 error: " + " ".join(f"{x:.1e}" for x in e[[0, 1, 2, 4, 9, 19, 29, 39, min(49, n - 1)]]))
    print("   first picks", rb.picks[:12])
# the same builders on the training block held in factored form (interface vectors only)
from romhighcontrast_amd import factored
fem = sm._fem
if fem.expansion_is_linear:
    K = fem.reduced_stride
    Y = ctx.alloc(M * K)
    _, t = T(lambda: (fem.solve_reduced(ctx.upload(a.reshape(M, -1)), M, Y), ctx.solve_status()))
    print(f"reduced solves (interface vectors, K = {K}): {t*1e3:.2f} ms")
    fs = factored.FactoredSnapshots(sm, Y, M)
    _, t = T(lambda: fs.map.build(3))
    print(f"energy map of the FE space (once): {t:.2f} s, ranks = {fs.map.build(3)}")
    h1f, t = T(lambda: factored.h10norm_factored(fs))
    print(f"H10 norms from the interface vectors: {t*1e3:.2f} ms, max rel diff to the stencil norms {np.abs(h1f / h1 - 1).max():.1e}")
    for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
        rbf = RB.ReducedBasisGreedy(mode)
        _, t = T(lambda: rbf.build(n, sm, fs, a, h1))
        rbr = RB.ReducedBasisGreedy(mode).build(n, sm, U, a, h1)
        same = sum(p == q for p, q in zip(rbf.picks, rbr.picks))
        e, er = np.array(rbf.max_errors), np.array(rbr.max_errors)
        print(f"factored greedy {mode}: n={n} in {t:.3f} s; picks equal to the row-based build: {same}/{n}; "
              f"max |error - error_rows| = {np.abs(e - er).max():.1e}; last error {e[-1]:.2e}")
