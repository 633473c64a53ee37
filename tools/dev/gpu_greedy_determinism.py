"""Are the greedy error curves (rows and factored block, both modes) the same bits from call to call? (dev probe)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import bench
from romhighcontrast_amd import _ffi, factored
from romhighcontrast_amd.lib import ReducedBasis as RB
from romhighcontrast_amd.lib.SolutionsManagers import SolutionsManagerFEM
from romhighcontrast_amd.lib import SolutionsManagers as SM_
blocks, N, M, n = (3, 3), 171, 1024, 50
sm = SolutionsManagerFEM(blocks, N)
ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
a = bench.workload_parameters("c4", blocks, M)
Ud = sm.generate_solutions_device(a)
h1 = sm.H10norm(Ud)
Yf = ctx.alloc(M * fem.reduced_stride)
fem.solve_reduced(ctx.upload(a.reshape(M, -1)), M, Yf)
ctx.solve_status()
fs = factored.FactoredSnapshots(sm, Yf, M)
for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
    er, ef = [], []
    for rep in range(3):
        er.append(np.array(RB.ReducedBasisGreedy(mode).build(n, sm, SM_.DeviceArray(Ud.buf, M, dim), a, h1).max_errors))   # (a plain block: the row route)
        ef.append(np.array(RB.ReducedBasisGreedy(mode).build(n, sm, fs, a, h1).max_errors))
    print(mode, "rows: calls identical", all(np.array_equal(er[0], e) for e in er), " factored: calls identical", all(np.array_equal(ef[0], e) for e in ef),
          " max |rows - factored|", np.abs(er[0] - ef[0]).max(), "at n =", int(np.argmax(np.abs(er[0] - ef[0]))) + 1,
          " rows spread", max(np.abs(er[0] - e).max() for e in er), " factored spread", max(np.abs(ef[0] - e).max() for e in ef))
    d = np.abs(er[0] - ef[0])
    print("   |rows - factored| per n (1e-9):", np.round(d * 1e9, 2).tolist())
