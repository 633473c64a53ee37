"""Does the step time depend on WHERE the snapshot block lands in HBM?  One process, one FE space: the same sweep into
differently placed U buffers (a pad allocation of varying size in front of each), 300 steps each."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from romhighcontrast_amd import _ffi

ctx = _ffi.get_context()
blocks, N, M = (2, 2), 128, 1024
fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
a_dev = ctx.upload(np.ascontiguousarray(bench.workload_parameters("c2", blocks, M)))
K = 300
def run(U, off=0):
    for _ in range(20):
        fem.solve_batch(a_dev, M, U, wait=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fem.solve_batch(a_dev, M, U, wait=False)
    ctx.solve_status()
    return (time.perf_counter() - t0) / K * 1e3
keep = []
for i, pad_mb in enumerate((0, 1, 33, 64, 100, 257, 512, 0, 0)):
    if pad_mb:
        keep.append(ctx.alloc(pad_mb * 131072 + 17 * i))   # (kept alive: the next allocation cannot reuse its place)
    U = ctx.alloc(M * fem.dim + 1024)
    t = [run(U) for _ in range(3)]
    print(f"pad {pad_mb:4d} MB: {t[0]:.4f} {t[1]:.4f} {t[2]:.4f} ms per step", flush=True)
    keep.append(U)
