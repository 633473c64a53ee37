"""Do the reduced solves of step s + 1 (k_solve1, latency bound) hide under the extension of step s (k_extend128)?
Feasibility probe with two contexts (= two compute streams) and no data dependency between them."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from romhighcontrast_amd import _ffi

blocks, N, M = (2, 2), 128, 1024
a = bench.workload_parameters("c2", blocks, M)
cs, fs, As, Ys, Us = [], [], [], [], []
for i in range(2):
    c = _ffi.Context(0); f = _ffi.Fem(c, blocks[0], blocks[1], N)
    cs.append(c); fs.append(f)
    As.append(c.upload(np.ascontiguousarray(a)))
    Ys.append(c.alloc(M * f.reduced_stride)); Us.append(c.alloc(M * f.dim))
    f.solve_reduced(As[i], M, Ys[i]); f.expand(As[i], M, Ys[i], Us[i]); c.synchronize()
K = 200
def run(what):
    for c in cs: c.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        if what in ("solve", "both", "serial"):
            fs[0].solve_reduced(As[0], M, Ys[0])
        if what in ("expand", "both"):
            fs[1].expand(As[1], M, Ys[1], Us[1])
        if what == "serial":
            fs[0].expand(As[0], M, Ys[0], Us[0])
    for c in cs: c.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for what in ("solve", "expand", "serial", "both", "both", "serial"):
    print(f"{what:8s} {run(what):.4f} ms per step", flush=True)
