"""What a caller of the reference API sees (dev probe): sm.generate_solutions(a) returns host rows, so the device->host
copy of the (M, dim) block is part of the call."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from romhighcontrast_amd.lib.SolutionsManagers import SolutionsManagerFEM
from romhighcontrast_amd import _ffi
ctx = _ffi.get_context(0)
sm = SolutionsManagerFEM((2, 2), 128)
M = 1024
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, 2, 2))
for rep in range(3):
    t0 = time.perf_counter(); U = sm.generate_solutions(a); t = time.perf_counter() - t0
    print(f"generate_solutions (host rows): {t*1e3:.1f} ms -> {M/t:.0f} solves/s")
Ud = sm.generate_solutions_device(a)
for rep in range(3):
    t0 = time.perf_counter(); Uh = Ud.numpy(); t = time.perf_counter() - t0
    print(f"download of {Uh.nbytes/1e6:.0f} MB: {t*1e3:.1f} ms -> {Uh.nbytes/t*1e-9:.1f} GB/s")
out = np.empty((M, sm.vspace_dim))
for rep in range(3):
    t0 = time.perf_counter(); Ud.buf.download_into(out) if hasattr(Ud.buf, "download_into") else None; t = time.perf_counter() - t0
    if hasattr(Ud.buf, "download_into"):
        print(f"download into an existing (touched) array: {t*1e3:.1f} ms -> {out.nbytes/t*1e-9:.1f} GB/s")
