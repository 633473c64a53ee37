"""Per-kernel HIP-event times (rom_profile_*) of the basis-stage calls: rom_pod at C2 / C3 / C5 size (`pod [M]`, `pod5`), rom_greedy at C4 size in both modes (`greedy`), the factored builders at C4 size (`factored`), or `all`.  Also the program the round-3 rocprofv3 --kernel-trace --stats summary of the basis stage was taken from (REPS=1 keeps it short)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from romhighcontrast_amd.lib import SolutionsManagers as SM, ReducedBasis as RB

def report(ctx, title, wall):
    rep = ctx.profile_report()
    tot = sum(v["total_ms"] for v in rep.values())
    print(f"== {title}: wall {wall*1e3:.2f} ms, kernels {tot:.2f} ms")
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"   {k:34s} {v['total_ms']:9.3f} ms  {v['launches']:5d} launches  {v['flops']/max(v['total_ms'],1e-9)*1e-9:8.2f} TF/s  {v['bytes']/max(v['total_ms'],1e-9)*1e-6:8.1f} GB/s")

which = sys.argv[1] if len(sys.argv) > 1 else "pod"
os.environ["ROMHC_PROF_DETAIL"] = "1"
if which in ("pod", "all"):
    sm = SM.SolutionsManagerFEM((2, 2), 128)
    ctx, dim = sm._ctx, sm.vspace_dim
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    a = bench.workload_parameters("c2", (2, 2), M)
    Ud = sm.generate_solutions_device(a)
    X = ctx.alloc(M * dim)
    for rep in range(3):
        X.copy_from(Ud.buf, M * dim)
        ctx.synchronize()
        if rep == 2:
            ctx.profile_reset(); ctx.profile(True)
        t0 = time.perf_counter()
        RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), 50)
        ctx.synchronize()
        w = time.perf_counter() - t0
        print("pod wall", w)
    ctx.profile(False)
    report(ctx, f"rom_pod {M} x {dim}", w)
if which in ("greedy", "all"):
    sm = SM.SolutionsManagerFEM((3, 3), 171)
    ctx, dim = sm._ctx, sm.vspace_dim
    M = 1024
    a = bench.workload_parameters("c4", (3, 3), M)
    Ud = sm.generate_solutions_device(a)
    h1 = sm.H10norm(Ud)
    for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
        for rep in range(2):
            ctx.synchronize()
            if rep == 1:
                ctx.profile_reset(); ctx.profile(True)
            t0 = time.perf_counter()
            RB.ReducedBasisGreedy(mode).build(50, sm, Ud, a, h1)
            ctx.synchronize()
            w = time.perf_counter() - t0
        ctx.profile(False)
        report(ctx, f"rom_greedy {mode} n=50", w)
if which in ("factored", "all"):
    # the same builders on the block held in factored form (rom_greedy_factored / rom_pod_factored, round 4)
    from romhighcontrast_amd import factored
    sm = SM.SolutionsManagerFEM((3, 3), 171)
    ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
    M = 1024
    a = bench.workload_parameters("c4", (3, 3), M)
    Yf = ctx.alloc(M * fem.reduced_stride)
    fem.solve_reduced(ctx.upload(a.reshape(M, -1)), M, Yf)
    ctx.solve_status()
    fs = factored.FactoredSnapshots(sm, Yf, M)
    t0 = time.perf_counter()
    ranks = fs.map.build(7)
    ctx.synchronize()
    print(f"rom_fem_energy_map(7) at 3x3 / N=171: {(time.perf_counter() - t0) * 1e3:.1f} ms, ranks {ranks}, Kc = {fs.map.Kc}")
    h1 = factored.h10norm_factored(fs)
    for mode in (RB.GREEDY_FOR_H10, RB.GREEDY_FOR_GALERKIN):
        for rep in range(2):
            ctx.synchronize()
            if rep == 1:
                ctx.profile_reset(); ctx.profile(True)
            t0 = time.perf_counter()
            RB.ReducedBasisGreedy(mode).build(50, sm, fs, a, h1)
            ctx.synchronize()
            w = time.perf_counter() - t0
        ctx.profile(False)
        report(ctx, f"rom_greedy_factored {mode} n=50", w)
    for rep in range(2):
        ctx.synchronize()
        if rep == 1:
            ctx.profile_reset(); ctx.profile(True)
        t0 = time.perf_counter()
        factored.pod_modes_factored(fs, 50)
        ctx.synchronize()
        w = time.perf_counter() - t0
    ctx.profile(False)
    report(ctx, "rom_pod_factored 1024 x 262144 (C4 block), 50 modes", w)
if which == "pod5":
    sm = SM.SolutionsManagerFEM((4, 4), 256)
    ctx, dim = sm._ctx, sm.vspace_dim
    M = 4096
    a = bench.workload_parameters("c5", (4, 4), M)
    Ud = sm.generate_solutions_device(a)
    X = ctx.alloc(M * dim)
    for rep in range(2):
        X.copy_from(Ud.buf, M * dim)
        ctx.synchronize()
        if rep == 1:
            ctx.profile_reset(); ctx.profile(True)
        t0 = time.perf_counter()
        RB.pod_modes(ctx, SM.DeviceArray(X, M, dim), 50)
        ctx.synchronize()
        w = time.perf_counter() - t0
        print("pod wall", w)
    ctx.profile(False)
    report(ctx, f"rom_pod {M} x {dim}", w)
