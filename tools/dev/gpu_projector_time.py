"""Wall time and per-kernel HIP-event times of the two projector calls (rom_project_h10, rom_galerkin_rom) at C4 size:
1024 snapshots of dim 262 144 onto the 50-vector greedy basis (dev probe)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB

os.environ["ROMHC_PROF_DETAIL"] = "1"
sm = SM.SolutionsManagerFEM((3, 3), 171)
ctx, dim = sm._ctx, sm.vspace_dim
M, n = 1024, 50
a = bench.workload_parameters("c4", (3, 3), M)
Ud = sm.generate_solutions_device(a)
rb = RB.ReducedBasisGreedy(RB.GREEDY_FOR_H10).build(n, sm, Ud, a, sm.H10norm(Ud))
basis = rb.basis if isinstance(rb.basis, SM.DeviceArray) else SM.DeviceArray(ctx.upload(np.ascontiguousarray(rb.basis)), n, dim)
for name, call in (("project_solutions", lambda: sm.project_solutions_device(Ud, basis)),
                   ("generate_fm_solutions", lambda: sm.generate_fm_solutions_device(a, basis))):
    for rep in range(3):
        ctx.synchronize()
        if rep == 2:
            ctx.profile_reset(); ctx.profile(True)
        t0 = time.perf_counter()
        out = call()
        ctx.synchronize()
        w = time.perf_counter() - t0
        del out
    ctx.profile(False)
    rep = ctx.profile_report()
    print(f"== {name}: wall {w*1e3:.2f} ms, kernels {sum(v['total_ms'] for v in rep.values()):.2f} ms")
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])[:10]:
        print(f"   {k:40s} {v['total_ms']:8.3f} ms {v['launches']:4d} launches {v['bytes']/max(v['total_ms'],1e-9)*1e-6:8.1f} GB/s")
