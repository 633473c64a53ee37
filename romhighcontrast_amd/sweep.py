"""Sharding of the parameter sweep over the GPUs of one node.

The reference's only parallelism is ``map`` over parameters through a process pool
(src/lib/SolutionsManagers.py:51,64-68).  Here rank g owns the contiguous rows
[g*Mp, min(M,(g+1)*Mp)), Mp = ceil(M/G); the shards are exchanged with ONE all-gather (RCCL over
xGMI on the GPU path) before the basis stage.  Because only the trailing shard(s) can be short, the
gathered (G*Mp, dim) block holds the M valid rows contiguously at the front.

What travels on the GPU path is the factored form of the shard: a snapshot row is a fixed linear image of
its system's interface vector (libromhc: rom_solve_reduced_async / rom_expand_batch_async), 1/85 of the
row at 256x256 / 2x2, so the ranks all-gather the interface vectors and every rank expands all of them
(bit-identical rows on every rank: the expansion is deterministic).  xGMI moves 6 MB per rank and step
instead of 528 MB.  `RcclSweep.generate_factored` stops there -- the gathered block stays in factored form,
which is all the POD needs (factored.py) -- `generate_solutions_device` also expands it on every rank.
"""
from __future__ import annotations

import os
import tempfile
import time

import numpy as np


def shard_rows(M: int, world: int) -> int:
    """Rows per rank (padded): ceil(M / world)."""
    return (int(M) + world - 1) // world if M > 0 else 0


def shard_bounds(M: int, world: int, rank: int):
    """Half-open row range owned by `rank`."""
    mp = shard_rows(M, world)
    lo = min(M, rank * mp)
    return lo, min(M, lo + mp)


def sharded_sweep(a_all, world: int, rank: int, solve_local, allgather, finish=None):
    """Run the sweep sharded: ``solve_local(a_shard, rows_padded)`` returns this rank's padded
    (Mp, width) block, ``allgather(block)`` returns the (world*Mp, width) concatenation in rank order,
    ``finish(gathered)`` (optional) turns the gathered block into snapshot rows when what was exchanged is
    their factored form.  Returns (gathered block, M): rows [0, M) are the snapshots in the order of ``a_all``."""
    a_all = np.asarray(a_all)
    M = a_all.shape[0]
    mp = shard_rows(M, world)
    lo, hi = shard_bounds(M, world, rank)
    local = solve_local(a_all[lo:hi], mp)
    full = allgather(local)
    return (finish(full) if finish is not None else full), M


# ---- the double-buffered step loop of the N > 1 sweep ---------------------------------------------------------------------
def run_step(backend, step_no: int, every: int = 1) -> int:
    """One step of the overlapped sweep; returns the buffer slot it used.  The exchange is issued once per GROUP of `every`
    consecutive steps (one larger all-gather instead of `every` small ones: a cross-stream dependency costs the compute
    stream ~25 us per step at C2, and xGMI prefers few large transfers); two buffer sets alternate between groups: the
    all-gather of group g (communication stream) overlaps the expansions of its last step and the whole of group g + 1
    (compute stream), so before a slot is rewritten the collective that last read it (group g - 2) must have finished.
    `backend`:
        wait_slot(k)            compute stream waits for the collective last issued with slot k (no-op if none)
        solve_local(k, j)       this rank's shard of the group's step j -> interface vectors into part j of the send
                                buffer of slot k (a short shard is padded)
        allgather_async(k, n)   parts 0 .. n-1 of slot k's send buffer -> its gathered buffer, not blocking the compute stream
        expand(k, j)            snapshot rows from the interface vectors of part j of slot k (own shard)
    bench.py drives GpuStepBackend with it; tests/test_host_logic.py drives a gloo stand-in through the SAME function
    (two processes, even and ragged M, slot reuse over several groups, every = 1 and 3)."""
    g, j = divmod(step_no, every)
    k = g & 1
    if j == 0:
        backend.wait_slot(k)
    backend.solve_local(k, j)
    if j == every - 1:
        backend.allgather_async(k, every)
    backend.expand(k, j)
    return k


def drain(backend, steps_done=None, every: int = 1):
    """End of a timed region: the shards of a last, incomplete group (steps_done % every of them) are sent, everything
    enqueued by run_step has finished, the last all-gather has landed."""
    if steps_done is not None and every > 1 and steps_done % every:
        backend.allgather_async((steps_done // every) & 1, steps_done % every)
    backend.drain()


class GpuStepBackend:
    """libromhc realisation of the step protocol: rom_solve_reduced_async / rom_comm_allgather_packed_async (pack + RCCL
    all-gather over xGMI, both on the context's communication stream) / rom_expand_batch_async, parameters resident in HBM.
    What travels is the COMPACT form of the interface vectors (the entries the expansion reads, 272 of 784 doubles per
    system at 256 x 256 / 2 x 2 -- the nodal part is recomputed by whoever expands), once per group of `every` steps.
    `M` rows per rank and step; a rank whose shard is short (`m_valid` < M) pads its send buffer with zero vectors once.
    Gathered layout of slot k after a collective of n parts: row (r * n + j) * M + m = rank r, part j, system m."""

    def __init__(self, ctx, fem, a_dev, M, world, *, U_loc=None, replicate=None, m_valid=None, every=1):
        self.ctx, self.fem, self.a_dev, self.M, self.world = ctx, fem, a_dev, int(M), int(world)
        self.every = int(every)
        assert self.every >= 1 and (replicate is None or self.every == 1), "the replicated row block is exchanged every step"
        self.m_valid = self.M if m_valid is None else int(m_valid)
        self.stride = fem.reduced_stride
        self.cstride = fem.compact_stride
        rows = self.every * self.M
        self.Y_loc = [ctx.alloc(max(rows * self.stride, 1)) for _ in range(2)]                   # full vectors of a group
        self.Yc_loc = [ctx.alloc(max(rows * self.cstride, 1)) for _ in range(2)]                 # packed: send buffers
        self.Yc_all = [ctx.alloc(max(self.world * rows * self.cstride, 1)) for _ in range(2)]    # gathered
        self.parts = [0, 0]                                                                       # parts in the last collective of a slot
        if self.m_valid < self.M:
            for y in self.Y_loc:
                y.fill(0.0)  # padding rows are gathered (and expanded by --replicate): they must hold finite numbers
        self.U_loc = U_loc if U_loc is not None else ctx.alloc(max(self.M * fem.dim, 1))
        self.replicate = replicate  # None, or (a_all_dev, U_all): also expand the gathered block on every rank
        self.Y_all_full = ctx.alloc(self.world * self.M * self.stride) if replicate is not None else None

    def wait_slot(self, k):
        self.ctx.comm_wait_slot(k)

    def solve_local(self, k, j=0):
        if self.m_valid:
            self.fem.solve_reduced(self.a_dev, self.m_valid, self.Y_loc[k], y_row0=j * self.M)

    def allgather_async(self, k, n=1):
        # pack + all-gather on the communication stream: the compute stream goes straight on to the expansion
        self.fem.allgather_packed_async(self.Y_loc[k], n * self.M, self.Yc_loc[k], self.Yc_all[k], 0, slot=k)
        self.parts[k] = n

    def gathered_vectors(self, k, rank, part, row0, rows, out=None):
        """Full-stride interface vectors of `rows` systems from `row0` of (rank, part) in the gathered buffer of slot k
        (the compute stream must already wait for the collective: comm_wait / wait_slot)."""
        out = out if out is not None else self.ctx.alloc(rows * self.stride)
        self.fem.unpack_reduced(self.Yc_all[k], rows, out, c_row0=(rank * self.parts[k] + part) * self.M + row0)
        return out

    def expand(self, k, j=0):
        if self.replicate is not None:
            a_all_dev, U_all = self.replicate
            self.ctx.comm_wait(False)                                       # compute stream waits for the gathered vectors
            self.fem.unpack_reduced(self.Yc_all[k], self.world * self.M, self.Y_all_full)
            self.fem.expand(a_all_dev, self.world * self.M, self.Y_all_full, U_all)   # the whole block as rows, on every rank
        elif self.m_valid:
            # rows of the own shard, while the vectors travel
            self.fem.expand(self.a_dev, self.m_valid, self.Y_loc[k], self.U_loc, y_row0=j * self.M)

    def drain(self):
        self.ctx.solve_status()  # waits for the compute stream; raises if any system was not positive definite
        self.ctx.comm_wait(True)


# ---- rendezvous for the RCCL unique id (one node, processes started by torch.distributed.run) ----
def _launcher_start_time() -> str:
    """Start time (clock ticks since boot) of the parent process = the launcher that spawned all ranks;
    together with its pid this identifies one launch even if pids and ports get reused."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def rendezvous_path() -> str:
    """File through which rank 0 publishes the RCCL unique id of ONE launch.  A launcher that NAMES its launch --
    ROMHC_LAUNCH_ID (bench.py's own launcher, a fresh uuid per launch) or a TORCHELASTIC_RUN_ID other than
    torch.distributed.run's default "none" (--rdzv-id) -- is identified by that name together with MASTER_ADDR /
    MASTER_PORT alone: the ranks may then be started through wrappers (their parent pids differ).  Without such a name
    the parent process (pid + start time) stands in for it: direct children of one launcher share it, and a later
    launch on the same port does not."""
    run_id = os.environ.get("ROMHC_LAUNCH_ID") or os.environ.get("TORCHELASTIC_RUN_ID")
    if run_id in (None, "", "none"):
        run_id = None
    tag = [os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0")]
    tag += [run_id] if run_id else ["none", str(os.getppid()), _launcher_start_time()]
    return os.path.join(tempfile.gettempdir(), "romhc_rdzv_" + "_".join(tag) + ".bin")


def _own_start_time() -> float:
    """Wall-clock time at which this process was started (seconds since the epoch)."""
    try:
        with open("/proc/self/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            uptime = float(f.read().split()[0])
        return time.time() - uptime + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, IndexError, ValueError):
        return 0.0


STALE_MARGIN_S = 30.0   # ranks of one launch start within seconds of each other; a file older than this is another launch's


def exchange_unique_id(rank: int, make_id, timeout_s: float = 120.0) -> bytes:
    """Rank 0 creates the id and publishes it atomically through a file unique to this launch
    (MASTER_ADDR/PORT + run id, or + parent pid and its start time); the other ranks poll for it.  A file written more
    than STALE_MARGIN_S before this process started is a leftover of a launch that died before its cleanup (same port,
    same name): it is ignored, rank 0 of THIS launch replaces it -- except under ROMHC_LAUNCH_ID, whose file name no other
    launch can have."""
    path = rendezvous_path()
    if rank == 0:
        uid = make_id()
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    # ROMHC_LAUNCH_ID is a fresh uuid per launch (bench.py's launcher): no other launch ever wrote this file, so its age
    # says nothing -- a rank that a wrapper starts a minute after rank 0 must still take it.  A run id the
    # user chose (--rdzv-id) or the parent-process fallback can meet a leftover of an earlier launch: there the age counts.
    not_before = 0.0 if os.environ.get("ROMHC_LAUNCH_ID") else _own_start_time() - STALE_MARGIN_S
    while time.time() - t0 < timeout_s:
        try:
            if os.stat(path).st_mtime >= not_before:
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) >= 128:
                    return uid[:128]
        except FileNotFoundError:
            pass
        time.sleep(0.02)
    raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout_s}s")


def cleanup_rendezvous(rank: int):
    if rank == 0:
        try:
            os.remove(rendezvous_path())
        except OSError:
            pass


class RcclSweep:
    """GPU realisation: the interface vectors of the local shard are computed by libromhc, exchanged with
    ncclAllGather, and expanded into the full snapshot block on every rank."""

    def __init__(self, sm, rank: int, world: int):
        self.sm, self.rank, self.world = sm, rank, world
        self.ctx = sm._ctx

    def _gather_interface_vectors(self, a_all):
        """(Y_all buffer, rows = world*Mp, M, a padded to `rows`): every rank's interface vectors, all-gathered."""
        ctx, fem = self.ctx, self.sm._fem
        stride = fem.reduced_stride
        a_all = np.ascontiguousarray(np.asarray(a_all, dtype=np.float64).reshape(len(a_all), -1))
        M = a_all.shape[0]
        mp = shard_rows(M, self.world)

        def solve_local(a_shard, mp):
            Y = ctx.alloc(max(mp * stride, 1))
            if len(a_shard) < mp:
                Y.fill(0.0)  # padding rows are expanded too (and ignored): they must hold finite numbers
            if len(a_shard):
                fem.solve_reduced(ctx.upload(a_shard), len(a_shard), Y)
            return Y

        def allgather(Y):
            # the compact form travels (the nodal part of a vector is an output of the expansion).  One rank takes the same
            # path -- pack, the collective whenever the context has a communicator, unpack -- so that a single-GPU run
            # exercises what every rank of an N-GPU job runs (world = 1 used to return Y untouched)
            kc = fem.compact_stride
            Yc, Yc_all, full = ctx.alloc(mp * kc), ctx.alloc(self.world * mp * kc), ctx.alloc(self.world * mp * stride)
            fem.pack_reduced(Y, mp, Yc)
            if self.world > 1 or ctx.has_comm:
                ctx.allgather(Yc, 0, Yc_all, 0, mp * kc)
            else:
                Yc_all.copy_from(Yc, mp * kc)
            fem.unpack_reduced(Yc_all, self.world * mp, full)
            return full

        Yall, M = sharded_sweep(a_all, self.world, self.rank, solve_local, allgather)
        rows = self.world * mp
        a_pad = np.ones((rows, a_all.shape[1]))
        a_pad[:M] = a_all  # the rows behind the M valid ones belong to short shards: any positive coefficient
        return Yall, rows, M, a_pad

    def generate_factored(self, a_all):
        """The gathered sweep in factored form (no snapshot row is materialised): see factored.py."""
        from .factored import FactoredSnapshots
        Yall, rows, M, _ = self._gather_interface_vectors(a_all)
        self.ctx.solve_status()
        return FactoredSnapshots(self.sm, Yall, M)

    def generate_solutions_device(self, a_all):
        """The gathered sweep as snapshot rows, replicated on every rank."""
        from .lib.SolutionsManagers import DeviceArray
        dim, ctx, fem = self.sm.vspace_dim, self.ctx, self.sm._fem
        Yall, rows, M, a_pad = self._gather_interface_vectors(a_all)
        U = ctx.alloc(max(rows * dim, 1))
        if rows:
            fem.expand(ctx.upload(a_pad), rows, Yall, U)
        ctx.solve_status()
        return DeviceArray(U, M, dim)
