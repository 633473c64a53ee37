#!/usr/bin/env python3
"""bench.py -- snapshot-sweep throughput of the HIP hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: rom_solve_batch of the BASELINE config C2
workload (2x2 blocks, N=128 -> 256x256 cells, dim 65 025, 1024-parameter sweep, seeded
10**U(0,2) coefficients) with the parameters already resident in HBM and the (M, dim) fp64
snapshot block left in HBM.  With N > 1 (one process per GPU, started by torch.distributed.run)
every rank solves its own 1024-parameter shard (weak scaling: N=8 is config C3, 8192 parameters)
and the shards are exchanged with one RCCL all-gather per step, inside the timed region; the
all-gather of step k runs on a communication stream and overlaps the solves of step k+1 (double
buffered), the region ends when the last all-gather has landed.

The timed region is bracketed by a barrier + stream synchronisation on both sides (RCCL all-reduce
for N > 1), the reported time is the max over ranks, and rank 0 prints ONE JSON line.

Host side is plain Python + ctypes (no torch): the rendezvous for the RCCL unique id goes through
a launch-scoped file (romhighcontrast_amd/sweep.py).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

# multi-process GPU work on this platform needs dmabuf IPC (normally already exported by the launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X spec fp64 matrix peak; the instruction itself sustains 48.8
HBM_PEAK_GBS = 8000.0


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(blocks, N, a, budget_s=15.0):
    """The oracle's restatement of the reference's method='lsqsparse' path (stencil -> CSC ->
    scipy.sparse.linalg.spsolve, src/lib/SolutionsManagers.py:31) on one host core (the reference
    default num_cores=1), over a bounded sample of the same sweep."""
    from oracle import rom_oracle as ro
    g = ro.Geometry(blocks, N)
    B = ro.load_vector(g)
    t0 = time.perf_counter()
    n = 0
    while n < len(a) and (time.perf_counter() - t0 < budget_s or n < 4):
        ro.solve_one(g, a[n], B, "lsqsparse")
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"first {n} of the {len(a)} C2 parameters, oracle stencil->CSC->scipy spsolve (SuperLU), "
                      f"{dt:.1f} s on 1 of {os.cpu_count()} host cores ({_cpu_model()})"}


def _pool_init():
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass


def _pool_solve(args):
    from oracle import rom_oracle as ro
    blocks, N, rows = args
    g = ro.Geometry(blocks, N)
    B = ro.load_vector(g)
    for a in rows:
        ro.solve_one(g, a, B, "lsqsparse")
    return len(rows)


def cpu_baseline_pool(blocks, N, a, per_worker=40):
    """The same CPU path with the reference's num_cores > 1 semantics (a process pool over the parameters,
    src/lib/SolutionsManagers.py:51,64-68) on this job's share of the host cores.  Runs BEFORE the GPU is
    initialised: the pool forks."""
    import multiprocessing as mp
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    cores = max(1, min(16, share))  # a one-GPU job's CPU share on the bench box
    n = min(len(a), cores * per_worker)
    chunks = [(blocks, N, a[i::cores][: (n + cores - 1) // cores]) for i in range(cores)]
    n = sum(len(c[2]) for c in chunks)
    with mp.get_context("fork").Pool(cores, initializer=_pool_init) as pool:
        pool.map(_pool_solve, [(blocks, N, a[:1])] * cores)  # imports + symbolic warm-up outside the timed region
        t0 = time.perf_counter()
        pool.map(_pool_solve, chunks, chunksize=1)
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"{n} of the {len(a)} C2 parameters over a pool of {cores} processes (one BLAS thread each), same "
                      f"oracle path, {dt:.1f} s; host has {os.cpu_count()} cores, {share} usable by this job"}


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on the C-level stdout at init; keep stdout for the one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--M", type=int, default=1024, help="parameters per GPU per step")
    ap.add_argument("--N", type=int, default=128, help="cells per block per dimension")
    ap.add_argument("--blocks", type=int, nargs=2, default=[2, 2])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pod", action="store_true")
    ap.add_argument("--no-pod-c3", action="store_true", help="skip the POD of the 8192-snapshot C3 block")
    ap.add_argument("--force-comm", action="store_true",
                    help="rehearsal: run the N>1 code path (RCCL communicator, all-gather of the interface vectors, "
                         "expansion of the gathered block) with one rank")
    ap.add_argument("--replicate", action="store_true",
                    help="N>1: also expand the gathered factored block into snapshot rows on every rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    pool_baseline = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.no_cpu_baseline and under_profiler:
        # the profiler's preloaded library has initialised the GPU already: no fork()ing process pool in that case
        pool_baseline = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                         "sample": "skipped: running under rocprofv3 (the pool forks, which must precede GPU initialisation)"}
    elif world == 1 and not args.no_cpu_baseline:
        # all-cores CPU figure first: it forks, which must happen before anything touches the GPU
        a0 = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(args.M,) + tuple(args.blocks))
        pool_baseline = cpu_baseline_pool(tuple(args.blocks), args.N, a0)

    from romhighcontrast_amd import _ffi, sweep
    from romhighcontrast_amd.lib.SolutionsManagers import DeviceArray, SolutionsManagerFEM
    from romhighcontrast_amd.lib.ReducedBasis import pod_modes

    # one process per GPU: device = LOCAL_RANK (ROMHC_FORCE_DEVICE only for rehearsing the launch
    # path on a box with fewer GPUs than ranks)
    import ctypes
    ndev = ctypes.c_int(0)
    _ffi.check(_ffi.load_library().rom_device_count(ctypes.byref(ndev)))
    # normally every GPU of the node is visible and LOCAL_RANK picks one; if the launcher pinned one
    # GPU per process (HIP_VISIBLE_DEVICES) only device 0 exists
    dev = int(os.environ.get("ROMHC_FORCE_DEVICE", local_rank % max(ndev.value, 1)))
    ctx = _ffi.get_context(dev)
    blocks, N, M = tuple(args.blocks), args.N, args.M
    sm = SolutionsManagerFEM(blocks, N, device=dev)
    fem, dim = sm._fem, sm.vspace_dim

    comm = world > 1 or args.force_comm
    if comm:
        with stdout_to_stderr():
            uid = sweep.exchange_unique_id(rank, ctx.comm_unique_id)
            ctx.comm_init(uid, rank, world)
            ctx.allreduce_host([0.0], "sum")  # first collective: connection set-up, banner

    def barrier():
        ctx.synchronize()
        if comm:
            ctx.allreduce_host([0.0], "sum")

    # synthetic sweep of SURVEY.md 8(d): seeded, all blocks free, contrast <= 1e2; rank r owns rows
    # [r*M, (r+1)*M) of the (world*M)-parameter sweep
    rng = np.random.default_rng(20240807)
    a_all = 10.0 ** rng.uniform(0, 2, size=(world * M,) + blocks)
    a_loc = a_all[rank * M:(rank + 1) * M]
    a_dev = ctx.upload(a_loc.reshape(M, -1))
    U_loc = ctx.alloc(M * dim)
    # N > 1: what is exchanged is the factored form of the shard -- the interface vectors (fem.reduced_stride
    # doubles per system, 1/83 of a snapshot row at C2).  Every rank materialises the rows of its own shard (the
    # (world*M, dim) block is resident in HBM across the ranks) and holds the WHOLE block in factored form, which
    # is what the POD consumes (romhighcontrast_amd/factored.py).  --replicate also expands the gathered block
    # on every rank (each rank then writes world x 528 MB per step); DESIGN.md section 7.
    stride = fem.reduced_stride
    step_no = [0]
    if comm:
        # two buffer pairs: the all-gather of step k (communication stream) overlaps the expansion of step k and
        # the reduced solves of step k+1 (compute stream)
        Y_loc = [ctx.alloc(max(M * stride, 1)) for _ in range(2)]
        Y_all = [ctx.alloc(max(world * M * stride, 1)) for _ in range(2)]
        if args.replicate:
            U_all = ctx.alloc(world * M * dim)
            a_all_dev = ctx.upload(a_all.reshape(world * M, -1))

    def step():
        if not comm:
            fem.solve_batch(a_dev, M, U_loc, wait=False)  # enqueued only: no host round trip per step
            return
        k = step_no[0] & 1
        step_no[0] += 1
        ctx.comm_wait_slot(k)                                    # the all-gather that last read Y_loc[k] is done
        fem.solve_reduced(a_dev, M, Y_loc[k])                    # this rank's shard: interface vectors ...
        ctx.allgather_async(Y_loc[k], 0, Y_all[k], 0, M * stride, slot=k)   # ... all-gathered (RCCL over xGMI)
        if args.replicate:
            ctx.comm_wait(False)                                 # compute stream waits for the gathered vectors
            fem.expand(a_all_dev, world * M, Y_all[k], U_all)    # the whole block as rows, on every rank
        else:
            fem.expand(a_dev, M, Y_loc[k], U_loc)                # the rows of the own shard, while the vectors travel

    def drain():
        ctx.solve_status()  # waits for the compute stream; raises if any system was not positive definite
        if comm:
            ctx.comm_wait(True)

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    drain()  # the timed region ends when the last all-gather has landed
    ev_ms = ctx.timer_stop()
    ctx.synchronize()
    wall = time.perf_counter() - t0
    if comm:
        wall = float(ctx.allreduce_host([wall], "max")[0])
    barrier()

    # second pass of the same K steps with every kernel launch bracketed by a HIP-event pair on its
    # launch stream (the per-kernel durations behind `roofline` / `kernels`).  Kept out of the timed
    # region above: ~70 event records per step cost ~8 % at 3 ms per step.
    ctx.profile_reset()
    ctx.profile(True)
    tp = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    ctx.synchronize()
    wall_prof = time.perf_counter() - tp
    ctx.profile(False)
    barrier()

    if comm:  # the gathered factored block must reproduce a plain local sweep bit for bit, on every rank
        ref = ctx.alloc(M * dim)
        fem.solve_batch(a_dev, M, ref)
        probe = [0, (M // 2) * dim + 17, M * dim - 1]
        if args.replicate:
            for off in probe:
                assert U_all.download(1, offset=rank * M * dim + off)[0] == ref.download(1, offset=off)[0]
        else:
            peer = (rank + 1) % world  # expand a few rows of the NEXT rank's shard from the gathered vectors
            a_peer = ctx.upload(a_all[peer * M:peer * M + 4].reshape(4, -1))
            rows = ctx.alloc(4 * dim)
            fem.expand(a_peer, 4, Y_all[(step_no[0] - 1) & 1], rows, y_row0=peer * M)
            chk = ctx.alloc(4 * dim)
            fem.solve_batch(a_peer, 4, chk)
            assert np.array_equal(rows.download(), chk.download())
            for off in probe:
                assert U_loc.download(1, offset=off)[0] == ref.download(1, offset=off)[0]
    if rank != 0:
        if comm:
            ctx.comm_destroy()
        return

    prof = ctx.profile_report()
    work = fem.solve_work()
    solves = world * M * args.steps
    value = solves / wall
    kernels = {}
    for name, v in prof.items():
        if v["launches"] == 0:
            continue
        avg_ms = v["total_ms"] / v["launches"]
        kernels[name] = {"launches_per_step": v["launches"] / args.steps, "avg_ms": round(avg_ms, 5),
                         "ms_per_step": round(v["total_ms"] / args.steps, 4),
                         "tflops": round(v["flops"] / v["total_ms"] * 1e-9, 3) if v["total_ms"] > 0 else 0.0,
                         "gbs": round(v["bytes"] / v["total_ms"] * 1e-6, 1) if v["total_ms"] > 0 else 0.0}
    dom = max((k for k in prof if k != "rccl_allgather"), key=lambda k: prof[k]["total_ms"])
    d = prof[dom]
    achieved = d["flops"] / d["total_ms"] * 1e-9  # algorithmic flops of the launches / their event time
    roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": FP64_MATRIX_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP64_MATRIX_PEAK_TFLOPS, 4), "traffic": None,
                "avg_launch_ms": round(d["total_ms"] / d["launches"], 5),
                "flops_per_launch": d["flops"] / d["launches"],
                "store_bytes_per_launch": 8.0 * M * blocks[0] * blocks[1] * (N - 1) ** 2 if dom.startswith("extend") else None,
                "sustained_mfma_tflops": 49.0,
                "frac_of_sustained": round(achieved / 49.0, 4),
                "note": "ALGORITHMIC flops (no padding of K or of the tiles) of the fp64 MFMA kernel (v_mfma_f64_16x16x4_f64) "
                        "against the spec fp64 matrix rate (64 cycles per instruction).  A register-only loop of that "
                        "instruction sustains 49 TFLOP/s on this part (one per 100-104 cycles per SIMD; "
                        "profiles/r01_mfma_f64_peak_microbench_v2.txt): sustained_mfma_tflops / frac_of_sustained.  The same "
                        "launch writes the snapshot rows (store_bytes_per_launch); store stream alone: 0.13-0.18 ms "
                        "(tools/hbm_write_bw.hip); kernel with its stores disabled: 0.198 ms; in-kernel cycle stamps and what "
                        "was tried: DESIGN.md section 5 and section 9"}
    # PMC-measured memory-side traffic of the dominant kernel, if a summary of separate
    # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command is committed
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    kname = {"factor_panel": "k_factor_panel", "diag_update": "k_diag_update", "extend": "k_extend",
             "extend_lr": "k_extend128" if N - 1 >= 96 and M >= 128 else "k_extend", "solve1": "k_solve1",
             "diag_potrf": "k_diag_potrf", "diag_inverse": "k_diag_inverse", "backsolve": "k_backsolve"}.get(dom)
    if os.path.exists(pmc_path) and (blocks, N, M) == ((2, 2), 128, 1024):
        pmc = json.load(open(pmc_path))["kernels"].get(kname)
        if pmc:
            roofline["traffic"] = pmc["traffic_bytes_per_launch"]
            roofline["traffic_note"] = ("bytes per launch from profiles/r01_pmc_traffic.json: rocprofv3 --pmc "
                                        "FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes; includes "
                                        "Infinity-Cache hits")
    out = {
        "metric": "snapshot_solves_per_sec", "value": round(value, 1), "unit": "solves/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C{2 if world == 1 else 3}: {blocks[0]}x{blocks[1]} blocks, N={N} "
                               f"({blocks[0] * N}x{blocks[1] * N} cells, dim {dim}), {M}-parameter sweep per GPU "
                               f"({world * M} total), a=10**U(0,2) seed 20240807"
                               + ((", RCCL all-gather of the snapshot block in factored form (interface vectors) each "
                                   "step" + (" + expansion of the whole block on every rank" if args.replicate else
                                             "; rows of the own shard materialised")) if world > 1 else ""),
                   "blocks_geometry": list(blocks), "N": N, "dim": dim, "M_per_gpu": M, "M_total": world * M,
                   "parallelism": f"sweep sharded over {world} GPU(s)"},
        "event_ms_per_step": round(ev_ms / args.steps, 4),
        "profiled_pass_ms_per_step": round(wall_prof / args.steps * 1e3, 4),
        "algorithm": {"flops_per_solve": work["flops_own"], "hbm_bytes_per_solve": work["bytes_own"],
                      "canonical_banded_flops_per_solve": work["flops_banded"],
                      "canonical_banded_bytes_per_solve": work["bytes_banded"],
                      "canonical_banded_equiv_gbs": round(value * work["bytes_banded"] * 1e-9, 1)},
        "roofline": roofline, "kernels": kernels,
    }

    if world == 1 and not args.no_pod:
        # secondary figure of the metric: POD-SVD GF/s on the snapshot block just produced
        X = ctx.alloc(M * dim).copy_from(U_loc, M * dim)
        r = 50
        ctx.synchronize()
        t0 = time.perf_counter()
        comps, sig = pod_modes(ctx, DeviceArray(X, M, dim), r, center=True)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        X.copy_from(U_loc, M * dim)          # second, warm run (first one pays one-off kernel loads)
        ctx.synchronize()
        t0 = time.perf_counter()
        comps, sig = pod_modes(ctx, DeviceArray(X, M, dim), r, center=True)
        ctx.synchronize()
        dt = min(dt, time.perf_counter() - t0)
        # F_pod = Gram + lift + eigh; only the lower half of the Gram matrix is computed -> M (M+1) D
        f_pod = float(M) * (M + 1) * dim + 2.0 * r * M * dim + 10.0 * M ** 3
        passes = getattr(pod_modes, "last_gram_passes", 1)
        out["pod"] = {"gflops": round(f_pod / dt * 1e-9, 1), "seconds": round(dt, 4), "M": M, "dim": dim, "modes": r,
                      "F_pod": f_pod, "sigma_1": float(sig[0]), "resolved_modes": int((sig > 0).sum()),
                      "gram_passes": passes,
                      "executed_gram_tflops": round(passes * float(M) * (M + 1) * dim / dt * 1e-12, 2),
                      "note": "centre + Gram on MFMA (lower tiles only) + device subspace iteration + lift, with one "
                              "deflation pass (a second Gram) for the modes below the Gram noise floor; F_pod = "
                              "M (M+1) D + 2 r M D + 10 M^3 (SURVEY 8d, symmetric half, ONE Gram) over the wall time incl. "
                              "the download of the r modes; executed_gram_tflops counts the Gram flops actually done"}
        if fem.expansion_is_linear:
            # the same figure on the block held in factored form (interface vectors; no row is read)
            from romhighcontrast_amd import factored
            Yf = ctx.alloc(M * fem.reduced_stride)
            fem.solve_reduced(a_dev, M, Yf)
            ctx.solve_status()
            fs = factored.FactoredSnapshots(sm, Yf, M)
            dtf = 1e9
            for _ in range(2):
                ctx.synchronize()
                t0 = time.perf_counter()
                comps_f, sig_f = factored.pod_modes_factored(fs, r)
                ctx.synchronize()
                dtf = min(dtf, time.perf_counter() - t0)
            out["pod_factored"] = {"gflops": round(f_pod / dtf * 1e-9, 1), "seconds": round(dtf, 4), "modes": r,
                                   "sigma_1_rel_diff": float(abs(sig_f[0] / sig[0] - 1)),
                                   "note": "POD of the same block from its interface vectors (U = Y B^T, Gram = Y (B^T B) Y^T, "
                                           "romhighcontrast_amd/factored.py); same F_pod accounting, i.e. the flops of "
                                           "the row-based algorithm over this algorithm's wall time"}
        if not args.no_pod_c3:
            # the POD where the MFMA work dominates the fixed costs: the C3 snapshot block (8192 x 65025, 4.3 GB:
            # what the 8 GPUs of C3 hold after their all-gather) built and decomposed on this one GPU
            M3 = 8 * M
            a3 = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M3, blocks[0] * blocks[1]))
            U3 = ctx.alloc(M3 * dim)
            fem.solve_batch(ctx.upload(a3), M3, U3)
            X3 = ctx.alloc(M3 * dim)
            dt3 = 1e9
            for _ in range(2):
                X3.copy_from(U3, M3 * dim)
                ctx.synchronize()
                t0 = time.perf_counter()
                comps3, sig3 = pod_modes(ctx, DeviceArray(X3, M3, dim), r, center=True)
                ctx.synchronize()
                dt3 = min(dt3, time.perf_counter() - t0)
            f3 = float(M3) * (M3 + 1) * dim + 2.0 * r * M3 * dim + 10.0 * M3 ** 3
            passes3 = getattr(pod_modes, "last_gram_passes", 1)
            out["pod_c3"] = {"gflops": round(f3 / dt3 * 1e-9, 1), "seconds": round(dt3, 4), "M": M3, "dim": dim, "modes": r,
                             "F_pod": f3, "gflops_without_eigh_term": round((f3 - 10.0 * M3 ** 3) / dt3 * 1e-9, 1),
                             "gram_passes": passes3,
                             "executed_gram_tflops": round(passes3 * float(M3) * (M3 + 1) * dim / dt3 * 1e-12, 2),
                             "sigma_1": float(sig3[0]), "resolved_modes": int((sig3 > 0).sum()),
                             "note": "same algorithm and accounting on the 8192-snapshot block of config C3, generated "
                                     "and decomposed on one GPU; gflops_without_eigh_term drops the 10 M^3 of the formula "
                                     "(the subspace iteration does far less than a full eigh)"}
            del X3, U3
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(blocks, N, a_loc)
        out["cpu_baseline_all_cores"] = pool_baseline
        # POD on the host: numpy.linalg.svd(X - mean) on a subsample of the same block that fits a few
        # seconds, LAPACK threads as configured on the box; same F_pod accounting (symmetric half) for its size
        Ms = min(M, 192)
        Xs = U_loc.download(Ms * dim, shape=(Ms, dim))
        t0 = time.perf_counter()
        Xc = Xs - Xs.mean(axis=0)
        sv = np.linalg.svd(Xc, full_matrices=False)[1]
        dts = time.perf_counter() - t0
        fs_ = float(Ms) * (Ms + 1) * dim + 2.0 * min(50, Ms) * Ms * dim + 10.0 * Ms ** 3
        out["pod_cpu_baseline"] = {"gflops": round(fs_ / dts * 1e-9, 1), "seconds": round(dts, 3), "M": Ms, "dim": dim,
                                   "sigma_1": float(sv[0]), "kind": "reference call",
                                   "sample": f"numpy.linalg.svd of the centred first {Ms} snapshots (the SVD inside "
                                             f"sklearn PCA, src/lib/ReducedBasis.py:196), all LAPACK threads of the host"}
    if comm:
        ctx.comm_destroy()
        sweep.cleanup_rendezvous(rank)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
