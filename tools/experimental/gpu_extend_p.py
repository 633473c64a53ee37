"""The persistent extension kernel (experimental build only: make -C romhighcontrast_amd/csrc clean all EXPERIMENTAL=1)
against k_extend128 on the geometries of tests/test_gpu_parity.py::test_extension_tilings_agree: the rows must be
identical.  Dev check, not collected by pytest: the product build does not contain the kernel."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romhighcontrast_amd import _ffi

ctx = _ffi.get_context(0)
for blocks, N, M in [((2, 2), 128, 130), ((3, 3), 24, 140), ((2, 3), 40, 200), ((2, 2), 90, 128), ((1, 2), 9, 129),
                     ((3, 2), 171, 128), ((5, 4), 33, 128)]:
    a = 10.0 ** np.random.default_rng(N).uniform(0, 3, size=(M, blocks[0] * blocks[1]))
    ab = ctx.upload(a)
    out = {}
    for name, env in (("default", {}), ("p", {"ROMHC_EXT_P": "1", "ROMHC_EXT_FLAT": "0"}),
                      ("pflat", {"ROMHC_EXT_P": "1", "ROMHC_EXT_FLAT": "1"}), ("flat", {"ROMHC_EXT_FLAT": "1"})):
        os.environ.update(env)
        fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
        for k in env:
            del os.environ[k]
        U = ctx.alloc(M * fem.dim)
        U.fill(float("nan"))
        fem.solve_batch(ab, M, U)
        out[name] = U.download(shape=(M, fem.dim))
    same = {n: bool(np.array_equal(out[n], out["default"])) for n in ("p", "pflat", "flat")}
    print(blocks, N, M, same)
    assert all(same.values())
print("ok")
