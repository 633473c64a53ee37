#!/usr/bin/env python3
"""bench.py -- snapshot-sweep throughput of the HIP hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config c2|c4|c5]

One "step" = one pass of the hot path over one batch: rom_solve_batch of the workload's parameter sweep with the
parameters already resident in HBM and the (M, dim) fp64 snapshot block left in HBM.  Workloads:

    c2 (default)  2x2 blocks, N=128 -> 256x256 cells, dim 65 025, 1024 parameters, a = 10**U(0,2)   [BASELINE configs[1]]
    c4            3x3 blocks, N=171 -> 513x513 cells, dim 262 144, 1024-parameter training set, contrast 1e8,
                  + greedy reduced basis to n=50 (both modes)                                         [configs[3]]
    c5            4x4 blocks, N=256 -> 1024x1024 cells, dim 1 046 529, 4096 parameters, a = 10**U(0,3),
                  + POD of the 4096-snapshot block                                                     [configs[4]]

With N > 1 (one process per GPU) every rank solves its own M-parameter shard (weak scaling: c2 at N=8 is config C3,
8192 parameters) and the shards are exchanged by RCCL all-gathers inside the timed region: the compact interface vectors
of --exchange-every consecutive steps (default 8) travel in ONE collective on a communication stream, which overlaps the
following steps (double buffered groups); the region ends when the last all-gather has landed.

Launch: `python bench.py --gpus N` with no launcher environment starts the N ranks itself (fresh child processes
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; the parent never touches the GPU and never exec()s); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks are the launcher's.  Either
way rank 0 prints ONE JSON line.

The timed region (exactly K steps) is bracketed by a barrier + stream synchronisation on both sides (RCCL
all-reduce for N > 1) and the reported time is the max over ranks.  It is repeated `--repeats` times (default 5):
`value` / `ms_per_step` are those of the MEDIAN region, `repeats` lists all of them.

Host side is plain Python + ctypes (no torch): the rendezvous for the RCCL unique id goes through a launch-scoped
file (romhighcontrast_amd/sweep.py).
"""
import argparse
import contextlib
import json
import os
import signal
import socket
import subprocess
import sys
import time
import uuid

import numpy as np

# multi-process GPU work on this platform needs dmabuf IPC (normally already exported by the launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X spec fp64 matrix peak; a register-only loop sustains 71-73
HBM_PEAK_GBS = 8000.0
SEED = 20240807                 # seeds and generator (PCG64 via default_rng) are part of the contract

CONFIGS = {
    "c2": dict(blocks=(2, 2), N=128, M=1024, label="C2", pool_per_worker=40, cpu_budget_s=15.0),
    "c4": dict(blocks=(3, 3), N=171, M=1024, label="C4", pool_per_worker=3, cpu_budget_s=15.0),
    "c5": dict(blocks=(4, 4), N=256, M=4096, label="C5", pool_per_worker=1, cpu_budget_s=20.0),
}


def workload_parameters(config: str, blocks, M_total: int) -> np.ndarray:
    """The synthetic sweep of SURVEY.md 8(d) for `config`, (M_total, nrb, ncb)."""
    rng = np.random.default_rng(SEED)
    if config == "c4":
        # row 0 = all ones; rows 1..k = ones with block j at 1e8 (the "limit solutions" the reference seeds its training
        # sets with, src/experiments/HighContrast.py:108,113); row k+1 = all 1e8; the rest 10**U(0,8)
        k = blocks[0] * blocks[1]
        a = np.ones((M_total,) + tuple(blocks))
        for j in range(min(k, M_total - 1)):
            a[1 + j].flat[j] = 1e8
        if M_total > k + 1:
            a[k + 1] = 1e8
        if M_total > k + 2:
            a[k + 2:] = 10.0 ** rng.uniform(0, 8, size=(M_total - k - 2,) + tuple(blocks))
        return a
    hi = 3 if config == "c5" else 2
    return 10.0 ** rng.uniform(0, hi, size=(M_total,) + tuple(blocks))


# =====================================================================================================================
# launcher: `python bench.py --gpus N` without a launcher environment starts its own ranks
# =====================================================================================================================
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv, child=None, poll_s: float = 0.05) -> int:
    """Start `n` fresh child processes (one rank per GPU) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment and wait for them.  The parent must not have touched the GPU (it has not: nothing of libromhc is
    loaded here) and never exec()s.  Rank 0 inherits stdout (the one JSON line); the other ranks' stdout goes to
    stderr.  If a rank fails, the remaining ranks -- exactly the PIDs started here -- are terminated and its exit
    code is returned; 0 only if every rank exited 0."""
    cmd = list(child) if child is not None else [sys.executable, os.path.abspath(__file__)]
    port, launch_id = _free_port(), uuid.uuid4().hex
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ROMHC_LAUNCH_ID=launch_id,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = dict(enumerate(procs))
    while live and rc == 0:
        for r, p in list(live.items()):
            code = p.poll()
            if code is None:
                continue
            del live[r]
            if code != 0:
                print(f"bench.py launcher: rank {r} (pid {p.pid}) exited with {code}; stopping the other ranks",
                      file=sys.stderr)
                rc = code if code > 0 else 1
                break
        if live and rc == 0:
            time.sleep(poll_s)
    for p in live.values():  # only after a failure: stop the ranks we started (by PID)
        with contextlib.suppress(ProcessLookupError):
            p.send_signal(signal.SIGTERM)
    t_end = time.time() + 10.0
    for p in live.values():
        try:
            p.wait(timeout=max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


# =====================================================================================================================
# CPU baselines (the oracle's restatement of the reference path; test infrastructure used as the checker only)
# =====================================================================================================================
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(blocks, N, a, budget_s=15.0, label="C2", min_solves=4):
    """The oracle's restatement of the reference's method='lsqsparse' path (stencil -> CSC ->
    scipy.sparse.linalg.spsolve, src/lib/SolutionsManagers.py:31) on one host core (the reference
    default num_cores=1), over a bounded sample of the same sweep."""
    from oracle import rom_oracle as ro
    g = ro.Geometry(blocks, N)
    B = ro.load_vector(g)
    t0 = time.perf_counter()
    n = 0
    while n < len(a) and (time.perf_counter() - t0 < budget_s or n < min_solves):
        ro.solve_one(g, a[n], B, "lsqsparse")
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"first {n} of the {len(a)} {label} parameters, oracle stencil->CSC->scipy spsolve (SuperLU), "
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


def cpu_baseline_pool(blocks, N, a, per_worker=40, label="C2", max_cores=16):
    """The same CPU path with the reference's num_cores > 1 semantics (a process pool over the parameters,
    src/lib/SolutionsManagers.py:51,64-68).  max_cores = 16: a one-GPU job's CPU share on the bench box; None: every
    core this job may use (Pool(os.cpu_count())).  Runs BEFORE the GPU is initialised: the
    pool forks."""
    import multiprocessing as mp
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    cores = max(1, min(max_cores, share) if max_cores else share)
    n = min(len(a), cores * per_worker)
    chunks = [(blocks, N, a[i::cores][: (n + cores - 1) // cores]) for i in range(cores)]
    n = sum(len(c[2]) for c in chunks)
    with mp.get_context("fork").Pool(cores, initializer=_pool_init) as pool:
        pool.map(_pool_solve, [(blocks, N, a[:1])] * cores)  # imports + symbolic warm-up outside the timed region
        t0 = time.perf_counter()
        pool.map(_pool_solve, chunks, chunksize=1)
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"{n} of the {len(a)} {label} parameters over a pool of {cores} processes (one BLAS thread each), same "
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


def pick_device(local_rank: int, local_world: int, ndev: int) -> int:
    """One process per GPU.  Every GPU of the node visible -> device LOCAL_RANK; a launcher that pinned one GPU per
    process (HIP_/ROCR_/CUDA_VISIBLE_DEVICES) -> device 0.  More ranks than visible GPUs otherwise is refused: two ranks
    on one device make RCCL fail with 'invalid usage' (ROMHC_FORCE_DEVICE overrides for launch rehearsals only)."""
    if "ROMHC_FORCE_DEVICE" in os.environ:
        return int(os.environ["ROMHC_FORCE_DEVICE"])
    if ndev >= local_world:
        return local_rank
    # ranks started by launch_ranks() inherit the parent's device visibility unchanged: a visibility variable there
    # means "this job has that many GPUs", never "one GPU per process"
    pinned = "ROMHC_LAUNCH_ID" not in os.environ and any(
        os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    if ndev == 1 and pinned:
        return 0
    raise SystemExit(f"bench.py: {local_world} ranks on this node but only {ndev} GPU(s) visible: one process per GPU is "
                     "required (RCCL refuses two ranks on one device)")


def gathered_rows_ok(ctx, fem, be, a_all, rank, world, M, dim, k_last) -> bool:
    """A few rows of the NEXT rank's shard, expanded from the vectors the last collective of slot `k_last` gathered, against a
    plain local sweep of the same parameters (bit for bit)."""
    peer = (rank + 1) % world
    a_peer = ctx.upload(a_all[peer * M:peer * M + 4].reshape(4, -1))
    rows = ctx.alloc(4 * dim)
    fem.expand(a_peer, 4, be.gathered_vectors(k_last, peer, be.parts[k_last] - 1, 0, 4), rows)
    chk = ctx.alloc(4 * dim)
    fem.solve_batch(a_peer, 4, chk)
    return bool(np.array_equal(rows.download(), chk.download()))


def pod_accounting(M, dim, r):
    """Flop count of a POD in the Gram formulation (SURVEY 8d without its 10 M^3 'eigh' term): the symmetric half of ONE
    Gram matrix + the lift of r modes.  Since round 5 rom_pod takes that route only for slowly decaying spectra; otherwise it
    executes thin products (`executed_flops` in the record) and `gflops` is an EQUIVALENT rate on this count."""
    return float(M) * (M + 1) * dim + 2.0 * r * M * dim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="how many times the K-step timed region is run (median reported)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--M", type=int, default=None, help="parameters per GPU per step (default: the config's)")
    ap.add_argument("--N", type=int, default=None, help="cells per block per dimension (default: the config's)")
    ap.add_argument("--blocks", type=int, nargs=2, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", "--no-pod", dest="no_extras", action="store_true",
                    help="only the sweep line: no POD / greedy / API-rate legs")
    ap.add_argument("--no-pod-c3", action="store_true", help="c2: skip the POD of the 8192-snapshot C3 block")
    ap.add_argument("--preroll", type=float, default=0.25,
                    help="seconds of untimed steps in front of the first timed region (on top of --warmup): the part reaches "
                         "its steady clocks only after ~0.2 s of work, and a region of 100 C2 steps is 25 ms")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="c2 on one GPU: skip the compact C4 / C5 legs (sweep, dominant kernel, greedy n = 50, C5 POD)")
    ap.add_argument("--force-comm", action="store_true",
                    help="rehearsal: run the N>1 code path (RCCL communicator, all-gather of the interface vectors, "
                         "expansion of the gathered block) with one rank")
    ap.add_argument("--exchange-every", type=int, default=8,
                    help="N > 1: one all-gather per this many steps (the shards of a group travel together); 1 = every step")
    ap.add_argument("--replicate", action="store_true",
                    help="N>1: also expand the gathered factored block into snapshot rows on every rank (the literal "
                         "(M/G, dim) row block of SURVEY 8e, replicated)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing below this line has run: no GPU call yet.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); using {world}", file=sys.stderr)
        args.gpus = world

    cfg = CONFIGS[args.config]
    blocks = tuple(args.blocks) if args.blocks else cfg["blocks"]
    N = args.N or cfg["N"]
    M = args.M or cfg["M"]
    custom = (blocks, N, M) != (cfg["blocks"], cfg["N"], cfg["M"])
    label = cfg["label"] if not custom else f"custom({args.config})"
    if label == "C2" and world > 1:
        label = "C3" if world == 8 else f"C2 x {world} GPUs (C3 at 8)"

    a_all = workload_parameters(args.config, blocks, world * M)
    a_loc = a_all[rank * M:(rank + 1) * M]

    pool_baseline = pool_baseline_full = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.no_cpu_baseline and under_profiler:
        # the profiler's preloaded library has initialised the GPU already: no fork()ing process pool in that case
        pool_baseline = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                         "sample": "skipped: running under rocprofv3 (the pool forks, which must precede GPU initialisation)"}
    elif world == 1 and not args.no_cpu_baseline:
        # all-cores CPU figure first: it forks, which must happen before anything touches the GPU
        pool_baseline = cpu_baseline_pool(blocks, N, a_loc, per_worker=cfg["pool_per_worker"], label=label)
        # ... and with every core of the host (the contract's Pool(os.cpu_count())); a short sample per worker
        # (C2 only: 256 workers hold 256 SuperLU factorisations -- 50 MB each at C2, 1.2 GB each at C5)
        if args.config == "c2" and not custom:
            pool_baseline_full = cpu_baseline_pool(blocks, N, a_loc, per_worker=5, label=label, max_cores=None)

    from romhighcontrast_amd import _ffi, sweep
    from romhighcontrast_amd.lib.SolutionsManagers import DeviceArray, SolutionsManagerFEM

    import ctypes
    ndev = ctypes.c_int(0)
    _ffi.check(_ffi.load_library().rom_device_count(ctypes.byref(ndev)))
    dev = pick_device(local_rank, local_world, ndev.value)
    ctx = _ffi.get_context(dev)
    ctx.synchronize()
    t_setup = time.perf_counter()
    sm = SolutionsManagerFEM(blocks, N, device=dev)
    ctx.synchronize()
    setup_s = time.perf_counter() - t_setup  # rom_fem_create: parameter-independent tables of the FE space
    fem, dim = sm._fem, sm.vspace_dim

    comm = world > 1 or args.force_comm
    rccl_ranks = 1
    if comm:
        with stdout_to_stderr():
            uid = sweep.exchange_unique_id(rank, ctx.comm_unique_id)
            ctx.comm_init(uid, rank, world)
            rccl_ranks = int(round(float(ctx.allreduce_host([1.0], "sum")[0])))  # first collective: set-up, banner
        assert rccl_ranks == world, f"RCCL sees {rccl_ranks} ranks, the launcher started {world}"

    def barrier():
        ctx.synchronize()
        if comm:
            ctx.allreduce_host([0.0], "sum")

    a_dev = ctx.upload(a_loc.reshape(M, -1))
    U_loc = ctx.alloc(M * dim)
    # N > 1: what is exchanged is the factored form of the shard -- the interface vectors (fem.reduced_stride
    # doubles per system, 1/83 of a snapshot row at C2).  Every rank materialises the rows of its own shard (the
    # (world*M, dim) block is resident in HBM across the ranks) and holds the WHOLE block in factored form, which
    # is what the POD consumes (romhighcontrast_amd/factored.py).  --replicate also expands the gathered block
    # on every rank (each rank then writes world x 528 MB per step); DESIGN.md section 7.
    stride = fem.reduced_stride
    step_no, last_slot, every = [0], [0], 1
    if comm:
        # two buffer pairs: the all-gather of step k (communication stream) overlaps the expansion of step k and
        # the reduced solves of step k+1 (compute stream).  The loop itself is sweep.run_step -- the function the
        # 2-rank CPU rehearsal drives (tests/test_host_logic.py).
        replicate = None
        if args.replicate:
            U_all = ctx.alloc(world * M * dim)
            a_all_dev = ctx.upload(a_all.reshape(world * M, -1))
            replicate = (a_all_dev, U_all)
        every = 1 if args.replicate else max(1, int(args.exchange_every))
        be = sweep.GpuStepBackend(ctx, fem, a_dev, M, world, U_loc=U_loc, replicate=replicate, every=every)
        stride = be.cstride  # what travels: the compact interface vectors

    def step():
        if not comm:
            fem.solve_batch(a_dev, M, U_loc, wait=False)  # enqueued only: no host round trip per step
            return
        last_slot[0] = sweep.run_step(be, step_no[0], every)
        step_no[0] += 1

    def drain():
        if comm:
            sweep.drain(be, step_no[0], every)  # (sends the shards of a last, incomplete group)
            step_no[0] = -(-step_no[0] // every) * every  # the next region starts a fresh group
        else:
            ctx.solve_status()  # waits for the compute stream; raises if any system was not positive definite

    for _ in range(max(args.warmup, 1) if comm and not args.replicate else args.warmup):
        step()   # (the exchange pre-flight below compares gathered buffers: at least one group must have been sent)
    drain()
    exchange_note = None
    if comm and not args.replicate:
        # Pre-flight of the grouped exchange: the warm-up above has just sent full groups and, whenever
        # warmup % every != 0, a last partial one.  A peer's rows expanded from the gathered vectors must equal a plain local
        # sweep of the same parameters bit for bit; if they do not with groups of `every` steps, the run falls back to one
        # collective per step (and says so in its line) instead of timing a broken path.
        ok = gathered_rows_ok(ctx, fem, be, a_all, rank, world, M, dim, last_slot[0])
        ok_all = int(round(float(ctx.allreduce_host([1.0 if ok else 0.0], "sum")[0]))) == world
        if not ok_all and every > 1:
            exchange_note = f"grouped exchange (every {every} steps) failed its bit-identity pre-flight: fell back to every step"
            print("bench.py: " + exchange_note, file=sys.stderr)
            every = 1
            be = sweep.GpuStepBackend(ctx, fem, a_dev, M, world, U_loc=U_loc, replicate=None, every=1)
            step_no[0] = 0
            for _ in range(max(1, args.warmup)):
                step()
            drain()
            ok = gathered_rows_ok(ctx, fem, be, a_all, rank, world, M, dim, last_slot[0])
            ok_all = int(round(float(ctx.allreduce_host([1.0 if ok else 0.0], "sum")[0]))) == world
        assert ok_all, "the gathered interface vectors do not reproduce the peer's rows"
    # untimed, time-based pre-roll: `steps` / `warmup` stay what the caller passed; the regions below then agree to ~1 %
    # (BENCH_r03: five regions falling 0.296 -> 0.263 ms per step behind 5 warm-up steps = 1.4 ms of work)
    preroll_steps = 0
    if args.preroll > 0:
        t_pre = time.perf_counter()
        while True:
            for _ in range(max(1, every) * 4):
                step()
            drain()
            preroll_steps += max(1, every) * 4
            done = time.perf_counter() - t_pre >= args.preroll
            if comm:
                done = float(ctx.allreduce_host([1.0 if done else 0.0], "max")[0]) > 0.5  # (every rank takes the same decision)
            if done:
                break
    walls, evs = [], []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(args.steps):
            step()
        drain()  # the timed region ends when the last all-gather has landed
        ev_ms = ctx.timer_stop()
        ctx.synchronize()
        w = time.perf_counter() - t0
        if comm:
            w = float(ctx.allreduce_host([w], "max")[0])
        barrier()
        walls.append(w)
        evs.append(ev_ms)
    order = np.argsort(walls)
    med = int(order[len(order) // 2])
    wall, ev_ms = walls[med], evs[med]

    # one more pass of the same K steps with every kernel launch bracketed by a HIP-event pair on its
    # launch stream (the per-kernel durations behind `roofline` / `kernels`).  Kept out of the timed
    # regions above: ~70 event records per step cost ~8 % at 3 ms per step.
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
            # (the last step's shard is the last part of the last collective of its slot)
            assert gathered_rows_ok(ctx, fem, be, a_all, rank, world, M, dim, last_slot[0])
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
    nblk = blocks[0] * blocks[1]
    roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": FP64_MATRIX_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP64_MATRIX_PEAK_TFLOPS, 4), "traffic": None,
                "avg_launch_ms": round(d["total_ms"] / d["launches"], 5),
                "flops_per_launch": d["flops"] / d["launches"],
                "store_bytes_per_launch": 8.0 * M * nblk * (N - 1) ** 2 if dom.startswith("extend") else None,
                "sustained_mfma_tflops": 72.0,
                "frac_of_sustained": round(achieved / 72.0, 4),
                "store_floor_ms": round(8.0 * M * nblk * (N - 1) ** 2 / 6.0e9, 4) if dom.startswith("extend") else None,
                "note": "ALGORITHMIC flops (no padding of K or of the tiles) of the fp64 MFMA kernel (v_mfma_f64_16x16x4_f64) "
                        "against the spec fp64 matrix rate (64 cycles per instruction).  A register-only loop of that "
                        "instruction (16 accumulators per wave) sustains 71-73 TFLOP/s on this part, 60-65 with a 6 TB/s store "
                        "stream interleaved (profiles/r02_mfma_store_overlap_microbench.txt; round 1's 49 TFLOP/s came from a "
                        "probe whose loop the compiler had filled with accumulator moves): sustained_mfma_tflops / "
                        "frac_of_sustained.  The same launch writes the snapshot rows (store_bytes_per_launch; store_floor_ms = "
                        "that at the 6 TB/s a store-only kernel reaches); DESIGN.md section 5"}
    # PMC-measured memory-side traffic of the dominant kernel, if a summary of separate
    # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command is committed
    cfg_sizes = {"c2": ((2, 2), 128, 1024), "c4": ((3, 3), 171, 1024), "c5": ((4, 4), 256, 4096)}
    pmc_files = {"c2": ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"),
                 "c4": ("r05_pmc_traffic_c4.json", "r04_pmc_traffic_c4.json", "r03_pmc_traffic_c4.json", "r02_pmc_traffic_c4.json"),
                 "c5": ("r05_pmc_traffic_c5.json", "r04_pmc_traffic_c5.json", "r03_pmc_traffic_c5.json", "r02_pmc_traffic_c5.json")}
    for pmc_name in pmc_files.get(args.config, ()):
        pmc_path = os.path.join(ROOT, "profiles", pmc_name)
        if os.path.exists(pmc_path) and (blocks, N, M) == cfg_sizes[args.config]:
            allk = json.load(open(pmc_path))["kernels"]
            pmc = next((v for k, v in allk.items() if dom.startswith("extend") and k.startswith("k_extend")), None) \
                if dom.startswith("extend") else allk.get("k_" + dom)
            if pmc:
                # a number from a COMMITTED profile, not from this run: say which build it was taken on (hash of the kernel
                # sources, tools/pmc_summary.py) and withhold it when that is not the build running now
                taken_on, now = json.load(open(pmc_path)).get("csrc_sha16"), csrc_sha16()
                roofline["traffic_source"] = {"file": f"profiles/{pmc_name}", "csrc_sha16_of_profile": taken_on, "csrc_sha16_of_this_build": now,
                                              "same_build": taken_on == now}
                if taken_on == now:
                    roofline["traffic"] = pmc["traffic_bytes_per_launch"]
                else:
                    roofline["traffic_of_other_build"] = pmc["traffic_bytes_per_launch"]
                roofline["traffic_note"] = (f"bytes per launch from profiles/{pmc_name}: rocprofv3 --pmc FETCH_SIZE (x2, "
                                            "gfx950) + WRITE_SIZE, separate passes of this same command; includes Infinity-Cache "
                                            "hits; `traffic` is null when the profile was taken on other kernel sources")
                break
    ms_all = [w / args.steps * 1e3 for w in walls]
    out = {
        "metric": "snapshot_solves_per_sec", "value": round(value, 1), "unit": "solves/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{label}: {blocks[0]}x{blocks[1]} blocks, N={N} "
                               f"({blocks[0] * N}x{blocks[1] * N} cells, dim {dim}), {M}-parameter sweep per GPU "
                               f"({world * M} total), SURVEY 8d parameters (seed {SEED})"
                               + ((f", one RCCL all-gather per {every} step(s) of the snapshot block in FACTORED form: the compact interface "
                                   f"vectors (the entries the expansion reads), {stride} doubles = {stride * 8} B per system, {M * stride * 8 / 1e6:.2f} MB sent per "
                                   "rank" + (" + expansion of the whole block into rows on every rank (the literal 8e row "
                                             "block, replicated)" if args.replicate else
                                             "; rows of the own shard materialised, any other row reproducible bit for bit "
                                             "from the gathered vectors")) if comm else ""),
                   "config": args.config, "blocks_geometry": list(blocks), "N": N, "dim": dim, "M_per_gpu": M,
                   "M_total": world * M, "parallelism": f"sweep sharded over {world} GPU(s)"},
        "repeats": {"regions": len(walls), "ms_per_step": [round(x, 4) for x in ms_all], "min_ms_per_step": round(min(ms_all), 4),
                    "median_ms_per_step": round(ms_all[med], 4), "max_ms_per_step": round(max(ms_all), 4),
                    "note": "each region = exactly `steps` steps between barriers; value / ms_per_step are the median region's"},
        "preroll": {"seconds": args.preroll, "steps": preroll_steps,
                    "note": "untimed steps in front of the first region, on top of `warmup` (clocks settle after ~0.2 s of work)"},
        "event_ms_per_step": round(ev_ms / args.steps, 4),
        "profiled_pass_ms_per_step": round(wall_prof / args.steps * 1e3, 4),
        "rccl_ranks": rccl_ranks if comm else 0,
        "exchange": ({"what": "compact interface vectors (factored snapshot block; the nodal part is recomputed by the expansion)",
                      "doubles_per_system": stride, "full_interface_vector_doubles": fem.reduced_stride, "steps_per_collective": every,
                      "sent_bytes_per_rank_per_step": M * stride * 8, "received_bytes_per_rank_per_step": world * M * stride * 8,
                      "row_block_bytes_per_rank": M * dim * 8, "replicated_rows": bool(args.replicate),
                      "preflight": exchange_note or "a peer's rows expanded from the gathered vectors equal a local sweep bit for bit "
                                                    "(checked after the warm-up, whose last group is partial when warmup % every != 0, and again after the run)"} if comm else None),
        "setup": {"setup_s": round(setup_s, 4),
                  "solves_per_sec_incl_setup_one_sweep": round(M / (setup_s + wall / args.steps), 1),
                  "note": "setup_s = SolutionsManagerFEM(...) = rom_fem_create (parameter-independent tables, once per FE "
                          "space); NOT in `value` (SURVEY 8d: inputs resident in HBM); the second figure is what a caller "
                          "pays for ONE sweep on a fresh FE space"},
        "algorithm": {"flops_per_solve": work["flops_own"], "hbm_bytes_per_solve": work["bytes_own"],
                      "canonical_banded_flops_per_solve": work["flops_banded"],
                      "canonical_banded_bytes_per_solve": work["bytes_banded"],
                      "canonical_banded_equiv_gbs": round(value * work["bytes_banded"] * 1e-9, 1)},
        "roofline": roofline, "kernels": kernels,
        "kernels_note": "per-kernel HIP-event times of one extra pass of the same K steps with the sweep on ONE stream "
                        "(rom_profile_enable): the stream the timed regions behind `value` use as well since round 5 (ROMHC_STREAMS=2..4 "
                        "splits a sweep into concurrent sub-batches; no longer the default)",
    }

    if world == 1 and not args.no_extras:
        extras_api_rate(out, sm, a_loc, M, dim)
        if args.config == "c2":
            extras_pod_c2(out, args, ctx, sm, fem, a_dev, U_loc, M, dim, blocks)
        elif args.config == "c4":
            extras_greedy_c4(out, ctx, sm, fem, a_loc, a_dev, U_loc, M, dim)
        elif args.config == "c5":
            extras_pod_c5(out, ctx, sm, fem, a_dev, U_loc, M, dim)
    pod_key = "pod_c3" if "pod_c3" in out else ("pod" if "pod" in out else None)
    if pod_key:
        pr = out[pod_key]
        out["roofline"]["secondary"] = {
            "metric": "pod_svd_gflops", "kernel": "rom_pod (k_gram128 + subspace iteration + deflation / lift GEMMs)",
            "bound": "mfma", "achieved": round(pr["gflops"] * 1e-3, 2), "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(pr["gflops"] * 1e-3 / FP64_MATRIX_PEAK_TFLOPS, 4), "seconds": pr["seconds"],
            "workload": f"{pr['modes']}-mode POD of the {pr['M']} x {pr['dim']} snapshot block "
                        + ("(config C3's gathered block, generated and decomposed on this GPU)" if pod_key == "pod_c3" else f"({label})"),
            "note": "USEFUL flops (symmetric half of ONE Gram matrix M(M+1)dim + lift 2 r M dim; no eigh term) over the wall "
                    "time of ONE rom_pod call incl. the download of the modes; the executed flops and the Gram kernel's own "
                    "rate are in the `" + pod_key + "` record"}
        if pod_key == "pod_c3" and "pod" in out:
            p1 = out["pod"]  # the metric's own 1-GPU geometry: 1024 x 65 025
            out["roofline"]["secondary"]["at_1gpu_geometry"] = {
                "workload": f"{p1['modes']}-mode POD of the {p1['M']} x {p1['dim']} block (C2: what ONE GPU holds)",
                "achieved": round(p1["gflops"] * 1e-3, 2), "unit": "TFLOP/s", "frac": round(p1["gflops"] * 1e-3 / FP64_MATRIX_PEAK_TFLOPS, 4),
                "seconds": p1["seconds"], "resolved_modes": p1.get("resolved_modes"),
                "note": "same accounting; at this size the Gram kernel is a sixth of the call, the rest small dense problems and launches"}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(blocks, N, a_loc, budget_s=cfg["cpu_budget_s"], label=label,
                                           min_solves=2 if args.config == "c5" else 4)
        out["cpu_baseline_all_cores"] = pool_baseline
        if pool_baseline_full is not None:
            out["cpu_baseline_host_cores"] = pool_baseline_full
        if args.config == "c2":
            # POD on the host: numpy.linalg.svd(X - mean) on a subsample of the same block that fits a few
            # seconds, LAPACK threads as configured on the box; same useful-flop accounting for its size
            Ms = min(M, 192)
            Xs = U_loc.download(Ms * dim, shape=(Ms, dim))
            t0 = time.perf_counter()
            Xc = Xs - Xs.mean(axis=0)
            sv = np.linalg.svd(Xc, full_matrices=False)[1]
            dts = time.perf_counter() - t0
            out["pod_cpu_baseline"] = {"gflops": round(pod_accounting(Ms, dim, min(50, Ms)) / dts * 1e-9, 1),
                                       "seconds": round(dts, 3), "M": Ms, "dim": dim,
                                       "sigma_1": float(sv[0]), "kind": "reference call",
                                       "sample": f"numpy.linalg.svd of the centred first {Ms} snapshots (the SVD inside "
                                                 f"sklearn PCA, src/lib/ReducedBasis.py:196), all LAPACK threads of the host"}
    if world == 1 and not args.no_extras and args.config == "c2" and not custom and not args.no_other_configs:
        # the two other single-GPU workloads of SURVEY 8d, compact, in the SAME line (the driver runs `python bench.py`)
        U_loc.free()
        a_dev.free()
        for other in ("c4", "c5"):
            try:
                out[other] = other_config_leg(ctx, dev, other)
            except Exception as e:  # (a leg must never cost the headline line)
                out[other] = {"error": f"{type(e).__name__}: {e}"}
    if comm:
        ctx.comm_destroy()
        sweep.cleanup_rendezvous(rank)
    print(json.dumps(out))


# =====================================================================================================================
# secondary legs (one GPU, after the timed sweep)
# =====================================================================================================================
def csrc_sha16():
    """Hash of the sources of the SWEEP kernels (the files that decide what k_extend128 / k_solve1 / the tile Cholesky do and how
    they are launched): identifies the build a committed traffic profile was taken on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "romhighcontrast_amd", "csrc")
    for name in ("rom_fem_kernels.hip", "rom_fem_solve.hip", "rom_fem_setup.hip", "rom_fem_dev.h", "rom_mma.h", "romhc_internal.h"):
        h.update(name.encode())
        h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def _timed(ctx, f, reps=2):
    best, res = 1e30, None
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        res = f()
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    return res, best


def extras_api_rate(out, sm, a_loc, M, dim):
    """What a caller of the reference API sees: sm.generate_solutions(a) returns host rows (M, dim), so upload, sweep
    and the PCIe copy of the block are part of the call.  The first calls land in pageable memory; from the third
    request of a size on, libromhc's page-locked pool is used (_ffi._pinned_array)."""
    if M * dim * 8 > 8e9:  # a C5-size block through PCIe four times is not worth the bench time
        Ms = max(1, int(8e9 // (dim * 8)))
    else:
        Ms = M
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        U = sm.generate_solutions(a_loc[:Ms])
        times.append(time.perf_counter() - t0)
        del U
    out["api"] = {"call": "sm.generate_solutions(a) -> host ndarray", "rows": Ms, "bytes": Ms * dim * 8,
                  "seconds": [round(t, 4) for t in times],
                  "solves_per_sec_pageable": round(Ms / min(times[:2]), 1),
                  "solves_per_sec_pinned": round(Ms / min(times[2:]), 1),
                  "note": "PCIe-inclusive rates of the reference-shaped API (never `value`): calls 1-2 copy into fresh pageable "
                          "memory, calls 3+ into a page-locked block of libromhc's pool"}


def extras_pod_c2(out, args, ctx, sm, fem, a_dev, U_loc, M, dim, blocks):
    from romhighcontrast_amd.lib.ReducedBasis import pod_modes
    from romhighcontrast_amd.lib.SolutionsManagers import DeviceArray
    r = 50
    X = ctx.alloc(M * dim)

    def run_rows(Xbuf, src, Mx):
        Xbuf.copy_from(src, Mx * dim)
        ctx.synchronize()
        t0 = time.perf_counter()
        res = pod_modes(ctx, DeviceArray(Xbuf, Mx, dim), r, center=True)
        ctx.synchronize()
        return res, time.perf_counter() - t0

    def pod_record(Mx, dt, sig, info):
        useful = pod_accounting(Mx, dim, r)
        rec = {"gflops": round(useful / dt * 1e-9, 1), "seconds": round(dt, 4), "M": Mx, "dim": dim, "modes": r,
               "useful_flops": useful, "sigma_1": float(sig[0]), "resolved_modes": int((sig > 0).sum()),
               "gflops_survey_formula_with_10M3": round((useful + 10.0 * Mx ** 3) / dt * 1e-9, 1)}
        rec.update(info)
        if "executed_flops" in info:
            rec["executed_gflops"] = round(info["executed_flops"] / dt * 1e-9, 1)
        return rec

    dt = 1e30
    for _ in range(4):  # (the first run pays one-off kernel loads; best of the rest)
        (_, sig), dt2 = run_rows(X, U_loc, M)
        dt = min(dt, dt2)
    ctx.profile_reset()
    ctx.profile(True)
    run_rows(X, U_loc, M)
    ctx.profile(False)
    launches = int(sum(v["launches"] for v in ctx.profile_report().values()))
    out["pod"] = pod_record(M, dt, sig, dict(getattr(pod_modes, "last_info", {}), kernel_launches_profiled=launches,
                            note="gflops = the flop count of the Gram formulation (SURVEY 8d: symmetric half of ONE Gram matrix + lift of r "
                                 "modes, no eigh term) over the wall time of pod_modes incl. the download of the r modes: an "
                                 "EQUIVALENT rate -- rom_pod executes `executed_flops` (thin products of the sketch passes; a "
                                 "Gram matrix only when `gram_passes` = 1)"))
    if fem.expansion_is_linear:
        # the same POD on the block held in factored form (interface vectors; no row is read)
        from romhighcontrast_amd import factored
        Yf = ctx.alloc(M * fem.reduced_stride)
        fem.solve_reduced(a_dev, M, Yf)
        ctx.solve_status()
        fs = factored.FactoredSnapshots(sm, Yf, M)
        (_, sig_f), dtf = _timed(ctx, lambda: factored.pod_modes_factored(fs, r), reps=5)
        info = dict(getattr(factored.pod_modes_factored, "last_info", {}))
        out["pod_factored"] = {"seconds": round(dtf, 4), "modes": r, "M": M,
                               "sigma_1_rel_diff": float(abs(sig_f[0] / sig[0] - 1)),
                               "resolved_modes": int((sig_f > 0).sum()),
                               "row_equivalent_gflops": round(pod_accounting(M, dim, r) / dtf * 1e-9, 1), **info,
                               "note": "POD of the same block from its interface vectors (U = Y B^T; romhighcontrast_amd/"
                                       "factored.py); row_equivalent_gflops = useful flops of the ROW algorithm over this "
                                       "algorithm's time (a speed-up statement, not an MFMA rate)"}
    if not args.no_pod_c3:
        # the POD where the MFMA work dominates the fixed costs: the C3 snapshot block (8192 x 65025, 4.3 GB:
        # what the 8 GPUs of C3 hold after their all-gather) built and decomposed on this one GPU
        M3 = 8 * M
        a3 = workload_parameters("c2", blocks, M3).reshape(M3, -1)
        U3 = ctx.alloc(M3 * dim)
        fem.solve_batch(ctx.upload(a3), M3, U3)
        X3 = ctx.alloc(M3 * dim)
        # (best of 5 calls)
        dt3 = 1e30
        for _ in range(5):
            (_, sig3), dt3b = run_rows(X3, U3, M3)
            dt3 = min(dt3, dt3b)
        out["pod_c3"] = pod_record(M3, dt3, sig3, dict(getattr(pod_modes, "last_info", {}),
                                   note="same algorithm and accounting on the 8192-snapshot block of config C3, generated and "
                                        "decomposed on one GPU"))
        # the POD's dominant kernel against the matrix peak: the Gram matrix of the (already centred / deflated) block
        G3 = ctx.alloc(M3 * M3)
        ctx.gram(M3, dim, X3, 0, dim, G3, 0, M3)
        ctx.synchronize()
        ts = []
        for _ in range(3):
            ctx.timer_start()
            ctx.gram(M3, dim, X3, 0, dim, G3, 0, M3)
            ts.append(ctx.timer_stop())
        tg = float(np.median(ts))
        nt = (M3 + 127) // 128
        fl_exec = nt * (nt + 1) / 2 * 2.0 * 128 * 128 * dim
        out["pod_c3"]["gram_kernel"] = {"ms": round(tg, 3), "achieved": round(fl_exec / tg * 1e-9, 2), "peak": FP64_MATRIX_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": round(fl_exec / tg * 1e-9 / FP64_MATRIX_PEAK_TFLOPS, 4),
                                        "bound": "mfma", "note": "k_gram128 + finish (HIP events, median of 3), flops of the lower 128 x 128 tiles it computes"}
        del X3, U3, G3


def extras_greedy_c4(out, ctx, sm, fem, a_loc, a_dev, U_loc, M, dim, factored_too=True):
    """C4: greedy reduced basis to n=50 on the 1024-parameter training set, both modes (src/lib/ReducedBasis.py:112-139),
    on snapshot rows (what a caller of the reference API has) and on the factored block."""
    from romhighcontrast_amd.lib import ReducedBasis as RB
    from romhighcontrast_amd.lib.SolutionsManagers import DeviceArray
    n = 50
    Ud = DeviceArray(U_loc, M, dim)
    h1, t_h1 = _timed(ctx, lambda: sm.H10norm(Ud))
    rec = {"n": n, "M": M, "h10norm_ms": round(t_h1 * 1e3, 3), "h10norm_gbs": round(8.0 * M * dim / t_h1 * 1e-9, 1)}
    picks = {}
    for tag, mode in (("h10", RB.GREEDY_FOR_H10), ("galerkin", RB.GREEDY_FOR_GALERKIN)):
        # (best of 2, like the POD legs: the first call of a mode pays one-off kernel loads and workspace allocations)
        rb, t = _timed(ctx, lambda: RB.ReducedBasisGreedy(mode).build(n, sm, Ud, a_loc, h1), reps=2)
        picks[tag] = rb.picks
        e = np.array(rb.max_errors)
        # algorithmic work of the row build: per iteration at basis size m two GEMMs 2 m D M + norms 12 M D
        fl = sum(4.0 * m * dim * M + 12.0 * M * dim for m in range(n))
        rec[f"rows_{tag}"] = {"seconds": round(t, 4), "tflops_algorithmic": round(fl / t * 1e-12, 2),
                              "max_rel_error_at_n": {str(k): float(e[k - 1]) for k in (1, 2, 5, 10, 20, 30, 40, 50) if k <= n},
                              "first_picks": rb.picks[:10]}
    if fem.expansion_is_linear and factored_too:
        from romhighcontrast_amd import factored
        Yf = ctx.alloc(M * fem.reduced_stride)
        fem.solve_reduced(a_dev, M, Yf)
        ctx.solve_status()
        fs = factored.FactoredSnapshots(sm, Yf, M)
        _, t_e = _timed(ctx, lambda: fs.map.build(3), reps=1)
        rec["energy_map_once_s"] = round(t_e, 3)  # rom_fem_energy_map: H^1_0 geometry + Galerkin forms, once per FE space
        # the reference-shaped call on a block fresh from the sweep: build(n, sm, sm.generate_solutions_device(a), a, h1) -- the
        # block carries its interface vectors and the greedy takes them in both modes (lib/ReducedBasis.py)
        Ud_api = sm.generate_solutions_device(a_loc)
        for tag, mode in (("h10", RB.GREEDY_FOR_H10), ("galerkin", RB.GREEDY_FOR_GALERKIN)):
            rb, t = _timed(ctx, lambda: RB.ReducedBasisGreedy(mode).build(n, sm, Ud_api, a_loc, h1), reps=2)
            rec[f"api_{tag}"] = {"seconds": round(t, 4), "route": "interface vectors" if Ud_api.factored is not None else "rows",
                                 "picks_equal_to_rows": int(sum(p == q for p, q in zip(rb.picks, picks[tag])))}
        # (the Galerkin greedy kept on the rows by the caller: 1.9e-10 from the 80-bit truth where the factored form is 4.5e-10 and the
        # reference's arithmetic 6.5e-10)
        rb, t = _timed(ctx, lambda: RB.ReducedBasisGreedy(RB.GREEDY_FOR_GALERKIN).build(n, sm, Ud_api, a_loc, h1, galerkin_on_interface_vectors=False), reps=2)
        rec["api_galerkin_rows"] = {"seconds": round(t, 4), "route": "rows (build(..., galerkin_on_interface_vectors=False))",
                                    "picks_equal_to_rows": int(sum(p == q for p, q in zip(rb.picks, picks["galerkin"])))}
        del Ud_api
        # the reference's literal pattern: solutions = sm.generate_solutions(a) -- a HOST array -- then build(n, sm, solutions, a, h1):
        # the manager kept the interface vectors of the array it returned; build() uploads the rows (PCIe: most of the time below),
        # checks bit for bit on the device that they are still the image of those vectors and runs on them
        Uh = sm.generate_solutions(a_loc)
        for tag, mode in (("h10", RB.GREEDY_FOR_H10), ("galerkin", RB.GREEDY_FOR_GALERKIN)):
            rb, t = _timed(ctx, lambda: RB.ReducedBasisGreedy(mode).build(n, sm, Uh, a_loc, h1), reps=2)
            rec[f"api_host_{tag}"] = {"seconds": round(t, 4), "route": "host rows -> upload -> interface vectors (verified on the device)",
                                      "picks_equal_to_rows": int(sum(p == q for p, q in zip(rb.picks, picks[tag])))}
        del Uh
        for tag, mode in (("h10", RB.GREEDY_FOR_H10), ("galerkin", RB.GREEDY_FOR_GALERKIN)):
            rb, t = _timed(ctx, lambda: RB.ReducedBasisGreedy(mode).build(n, sm, fs, a_loc, h1), reps=2)
            rec[f"factored_{tag}"] = {"seconds": round(t, 4),
                                      "picks_equal_to_rows": int(sum(p == q for p, q in zip(rb.picks, picks[tag]))),
                                      "last_max_rel_error": float(rb.max_errors[-1])}
    out["greedy"] = rec


def other_config_leg(ctx, dev, config):
    """Compact run of another single-GPU workload (C4 / C5) for the default line: sweep rate (median of 3 regions behind a
    pre-roll), per-kernel HIP events of one pass, dominant kernel against its roofline, and the workload's basis stage
    (C4: greedy n = 50 in both modes on rows, on the factored block and through the plain build() call on a block fresh from
    the sweep; C5: 50-mode POD of the rows and of the interface vectors).  `python bench.py --config c4|c5` adds the CPU
    baselines and API rates."""
    from romhighcontrast_amd.lib.SolutionsManagers import SolutionsManagerFEM
    cfg = CONFIGS[config]
    blocks, N, M = cfg["blocks"], cfg["N"], cfg["M"]
    steps = 20 if config == "c4" else 4
    ctx.synchronize()
    t0 = time.perf_counter()
    sm = SolutionsManagerFEM(blocks, N, device=dev)
    ctx.synchronize()
    setup_s = time.perf_counter() - t0
    fem, dim = sm._fem, sm.vspace_dim
    a_loc = workload_parameters(config, blocks, M)
    a_dev = ctx.upload(a_loc.reshape(M, -1))
    U = ctx.alloc(M * dim)
    for _ in range(2):
        fem.solve_batch(a_dev, M, U, wait=False)
    ctx.solve_status()
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.2:
        fem.solve_batch(a_dev, M, U, wait=False)
        ctx.solve_status()
    walls = []
    for _ in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fem.solve_batch(a_dev, M, U, wait=False)
        ctx.solve_status()
        walls.append(time.perf_counter() - t0)
    wall = sorted(walls)[1]
    ctx.profile_reset()
    ctx.profile(True)
    for _ in range(steps):
        fem.solve_batch(a_dev, M, U, wait=False)
    ctx.solve_status()
    ctx.profile(False)
    prof = ctx.profile_report()
    kernels = {k: {"ms_per_step": round(v["total_ms"] / steps, 4), "launches_per_step": v["launches"] / steps,
                   "tflops": round(v["flops"] / v["total_ms"] * 1e-9, 2) if v["total_ms"] > 0 else 0.0,
                   "frac": round(v["flops"] / v["total_ms"] * 1e-9 / FP64_MATRIX_PEAK_TFLOPS, 4) if v["total_ms"] > 0 and v["flops"] > 0 else None}
               for k, v in prof.items() if v["launches"]}
    dom = max(prof, key=lambda k: prof[k]["total_ms"])
    d = prof[dom]
    ach = d["flops"] / d["total_ms"] * 1e-9
    rec = {"workload": f"{cfg['label']}: {blocks[0]}x{blocks[1]} blocks, N={N}, dim {dim}, {M}-parameter sweep (SURVEY 8d parameters, seed {SEED})",
           "value": round(M * steps / wall, 1), "unit": "solves/s", "steps": steps, "ms_per_step": round(wall / steps * 1e3, 4),
           "regions_ms_per_step": [round(w / steps * 1e3, 4) for w in walls], "setup_s": round(setup_s, 3),
           "roofline": {"kernel": dom, "bound": "mfma", "achieved": round(ach, 3), "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / FP64_MATRIX_PEAK_TFLOPS, 4), "avg_launch_ms": round(d["total_ms"] / d["launches"], 5)},
           "kernels": kernels,
           "note": "compact leg of the default run: the timed regions run the sweep as the library does by default (one stream), "
                   "`kernels` is one extra pass with every launch bracketed by HIP events"}
    sub = {}
    if config == "c4":
        extras_greedy_c4(sub, ctx, sm, fem, a_loc, a_dev, U, M, dim, factored_too=True)
        rec["greedy"] = sub["greedy"]
    else:
        extras_pod_c5(sub, ctx, sm, fem, a_dev, U, M, dim, factored_too=True)
        rec["pod"] = sub["pod"]
        if "pod_factored" in sub:
            rec["pod_factored"] = sub["pod_factored"]
        rec["pod"]["frac_of_matrix_peak"] = round(rec["pod"].get("executed_flops", 0.0) / rec["pod"]["seconds"] * 1e-12 / FP64_MATRIX_PEAK_TFLOPS, 4)
    del U, a_dev, sm, fem
    return rec


def extras_pod_c5(out, ctx, sm, fem, a_dev, U_loc, M, dim, factored_too=True):
    """C5: POD of the 4096 x 1 046 529 block (src/lib/ReducedBasis.py:189-200), rows and factored."""
    from romhighcontrast_amd.lib.ReducedBasis import pod_modes
    from romhighcontrast_amd.lib.SolutionsManagers import DeviceArray
    r = 50
    X = ctx.alloc(M * dim)
    dt = 1e30
    for _ in range(3):  # (best of 3: the first call pays one-off kernel loads)
        X.copy_from(U_loc, M * dim)
        ctx.synchronize()
        t0 = time.perf_counter()
        _, sig = pod_modes(ctx, DeviceArray(X, M, dim), r, center=True)
        ctx.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    useful = pod_accounting(M, dim, r)
    out["pod"] = {"gflops": round(useful / dt * 1e-9, 1), "seconds": round(dt, 4), "M": M, "dim": dim, "modes": r,
                  "useful_flops": useful, "sigma_1": float(sig[0]), "resolved_modes": int((sig > 0).sum()),
                  **getattr(pod_modes, "last_info", {}),
                  "note": "gflops = flop count of the Gram formulation (symmetric half of ONE Gram + lift, no eigh term) over the wall "
                          "time of pod_modes; executed: `executed_flops` (a Gram matrix when `gram_passes` = 1)"}
    del X
    if fem.expansion_is_linear and factored_too:
        from romhighcontrast_amd import factored
        Yf = ctx.alloc(M * fem.reduced_stride)
        fem.solve_reduced(a_dev, M, Yf)
        ctx.solve_status()
        fs = factored.FactoredSnapshots(sm, Yf, M)
        (_, sig_f), dtf = _timed(ctx, lambda: factored.pod_modes_factored(fs, r))
        out["pod_factored"] = {"seconds": round(dtf, 4), "modes": r, "M": M,
                               "sigma_rel_diff_max": float(np.abs(sig_f[sig > 1e-6 * sig[0]] / sig[sig > 1e-6 * sig[0]] - 1).max()),
                               "row_equivalent_gflops": round(useful / dtf * 1e-9, 1),
                               **getattr(factored.pod_modes_factored, "last_info", {})}


if __name__ == "__main__":
    main()
