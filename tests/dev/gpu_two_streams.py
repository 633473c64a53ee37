"""Does overlapping two sweeps (two contexts = two HIP streams) raise the throughput?  (dev probe)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd import _ffi
N, M, K = 128, 1024, 40
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 4))
ctxs = [_ffi.Context(0), _ffi.Context(0)]
fems = [_ffi.Fem(c, 2, 2, N) for c in ctxs]
abs_ = [c.upload(a) for c in ctxs]
Us = [c.alloc(M * f.dim) for c, f in zip(ctxs, fems)]
for mode in ("one", "two"):
    use = 1 if mode == "one" else 2
    for w in range(3):
        for i in range(use):
            fems[i].solve_batch(abs_[i], M, Us[i], wait=False)
    for i in range(use):
        ctxs[i].solve_status()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(K):
            i = k % use
            fems[i].solve_batch(abs_[i], M, Us[i], wait=False)
        for i in range(use):
            ctxs[i].solve_status()
        best = min(best, time.perf_counter() - t0)
    print(f"{mode} stream(s): {best / K * 1e3:.4f} ms/step -> {M * K / best:.0f} solves/s")
