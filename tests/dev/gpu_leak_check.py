"""Device memory after repeated create / use / destroy of FE spaces, blocks and builders (dev probe; GPU): free bytes from
hipMemGetInfo before and after N rounds."""
import os, sys, gc, ctypes, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB
from romhighcontrast_amd import factored, _ffi
logging.disable(logging.WARNING)
hip = ctypes.CDLL("libamdhip64.so")
def free_bytes():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value
ctx = _ffi.get_context()
rng = np.random.default_rng(0)
def one_round(k):
    blocks = [(2, 2), (3, 2), (1, 3)][k % 3]
    N = [16, 24, 32][k % 3]
    M = 64
    sm = SM.SolutionsManagerFEM(blocks, N)
    a = 10.0 ** rng.uniform(0, 2, size=(M,) + blocks)
    Ud = sm.generate_solutions_device(a)
    h1 = sm.H10norm(Ud)
    rb = RB.ReducedBasisGreedy(RB.GREEDY_FOR_H10).build(8, sm, Ud, a, h1)
    rb2 = RB.ReducedBasisGreedy(RB.GREEDY_FOR_GALERKIN).build(6, sm, Ud, a, h1)
    pca = RB.ReducedBasisPCA(add_inf_solutions=False).build(n=8, sm=sm, solutions2train=Ud, a2train=a)
    Q = RB.orthonormalize_base(np.asarray(rb.basis)[:4])   # (rb.basis holds the raw picked snapshots, like the reference's)
    P = sm.project_solutions(Ud, Q)
    F = sm.generate_fm_solutions(a, Q)
    X = ctx.alloc(M * sm.vspace_dim).copy_from(Ud.buf, M * sm.vspace_dim)
    RB.pod_modes(ctx, SM.DeviceArray(X, M, sm.vspace_dim), 10)
    del sm, Ud, rb, rb2, pca, X
for k in range(3):
    one_round(k)     # (warm the pools)
gc.collect(); ctx.synchronize()
f0 = free_bytes()
for k in range(30):
    one_round(k)
gc.collect(); ctx.synchronize()
f1 = free_bytes()
print(f"free before {f0 / 2**20:.1f} MiB, after 30 rounds {f1 / 2**20:.1f} MiB, difference {(f0 - f1) / 2**20:.2f} MiB")
