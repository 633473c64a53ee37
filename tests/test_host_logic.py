"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the product fails loudly without a GPU, and the sweep sharding (N > 1 path) is correct
under a real 2-process gloo group."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "romhc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rom_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from romhighcontrast_amd import _ffi
    lib = _ffi.load_library()  # dlopen only, no GPU call
    names = header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"libromhc.so lacks {n}"
        assert n in _ffi.PROTOTYPES, f"_ffi.PROTOTYPES lacks {n}"
    assert sorted(_ffi.PROTOTYPES) == names
    assert lib.rom_version() >= 100
    assert lib.rom_last_error() is not None


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: constructing the solver on a box without an MI355X must raise."""
    import ctypes
    from romhighcontrast_amd import _ffi
    lib = _ffi.load_library()
    n = ctypes.c_int(0)
    st = lib.rom_device_count(ctypes.byref(n))
    if st == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    from src.lib.SolutionsManagers import SolutionsManagerFEM
    with pytest.raises(_ffi.RomLibraryError):
        SolutionsManagerFEM((2, 2), 4)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "romhighcontrast_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle order", "").replace("// oracle order", ""), (dirpath, f)


def test_shard_bounds_cover_and_order():
    from romhighcontrast_amd import sweep
    for M in (0, 1, 5, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                lo, hi = sweep.shard_bounds(M, world, r)
                assert 0 <= lo <= hi <= M and hi - lo <= sweep.shard_rows(M, world)
                rows += list(range(lo, hi))
            assert rows == list(range(M))


_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from romhighcontrast_amd import sweep
from oracle import rom_oracle as ro     # tests may use the oracle as the compute stand-in

dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
g = ro.Geometry((2, 2), 6)
M = int(sys.argv[4])
a = 10.0 ** np.random.default_rng(7).uniform(0, 2, size=(M, 2, 2))

def solve_local(a_shard, mp):
    out = np.zeros((mp, g.dim))
    if len(a_shard):
        out[:len(a_shard)] = ro.generate_solutions(g, a_shard)
    return out

def allgather(local):
    t = torch.from_numpy(local)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return torch.cat(outs).numpy()

full, Mout = sweep.sharded_sweep(a, world, rank, solve_local, allgather)
ref = ro.generate_solutions(g, a)
assert Mout == M and full.shape[0] == world * sweep.shard_rows(M, world)
assert np.array_equal(full[:M], ref), "gathered rows differ from the unsharded sweep"
assert not full[M:].any()
# unique-id rendezvous through the launch-scoped file
os.environ["MASTER_PORT"] = sys.argv[2]
uid = sweep.exchange_unique_id(rank, lambda: bytes(range(128)))
assert uid == bytes(range(128))
dist.barrier()
sweep.cleanup_rendezvous(rank)
dist.destroy_process_group()
print("OK", rank)
'''


@pytest.mark.parametrize("M", [8, 7])
def test_sharded_sweep_two_ranks_gloo(tmp_path, M):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r), str(M)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "OK" in o


def test_rendezvous_key_and_stale_files(tmp_path, monkeypatch):
    """The RCCL unique-id file: torch.distributed.run's default run id "none" does not name a launch (the parent process
    does), a real run id or ROMHC_LAUNCH_ID does; a leftover of an earlier launch under the same name is not taken for
    this launch's id."""
    import tempfile
    import threading
    import time
    from romhighcontrast_amd import sweep
    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29999")
    monkeypatch.delenv("ROMHC_LAUNCH_ID", raising=False)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    by_parent = sweep.rendezvous_path()
    assert str(os.getppid()) in os.path.basename(by_parent)
    monkeypatch.delenv("TORCHELASTIC_RUN_ID")
    assert sweep.rendezvous_path() == by_parent
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job42")
    named = sweep.rendezvous_path()
    assert "job42" in named and str(os.getppid()) not in os.path.basename(named).split("job42")[1]
    monkeypatch.setenv("ROMHC_LAUNCH_ID", "abc")
    assert "abc" in sweep.rendezvous_path() and "job42" not in sweep.rendezvous_path()
    # a launch named by ROMHC_LAUNCH_ID (a fresh uuid per launch) takes its file whatever its age: a rank that a wrapper
    # starts long after rank 0 published must not time out
    path = sweep.rendezvous_path()
    with open(path, "wb") as f:
        f.write(bytes(range(128)))
    old = time.time() - 3600.0
    os.utime(path, (old, old))
    assert sweep.exchange_unique_id(1, None, timeout_s=2.0) == bytes(range(128))
    os.remove(path)
    # under a name the user chose (or the parent-process fallback) a stale file -- another launch's id, written long before
    # this process started -- is ignored until rank 0 replaces it
    monkeypatch.delenv("ROMHC_LAUNCH_ID")
    path = sweep.rendezvous_path()
    with open(path, "wb") as f:
        f.write(b"\xff" * 128)
    old = time.time() - 3600.0
    os.utime(path, (old, old))
    fresh = bytes(range(128))

    def rank0_later():
        time.sleep(0.3)
        sweep.exchange_unique_id(0, lambda: fresh)

    th = threading.Thread(target=rank0_later)
    th.start()
    got = sweep.exchange_unique_id(1, None, timeout_s=20.0)
    th.join()
    assert got == fresh
    sweep.cleanup_rendezvous(0)
    assert not os.path.exists(path)
    with pytest.raises(TimeoutError):
        sweep.exchange_unique_id(1, None, timeout_s=0.2)


_STEP_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from romhighcontrast_amd import sweep
from oracle import rom_oracle as ro     # the compute stand-in: interface vectors := the oracle's snapshot rows

dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
g = ro.Geometry((2, 2), 4)
M = int(sys.argv[4])                   # total sweep size; the last shard is short when M is odd
mp = sweep.shard_rows(M, world)
lo, hi = sweep.shard_bounds(M, world, rank)
a0 = 10.0 ** np.random.default_rng(7).uniform(0, 2, size=(M, 2, 2))


class GlooBackend:
    """Stand-in for GpuStepBackend: same protocol, NumPy buffers, gloo's ASYNCHRONOUS all-gather as the collective.
    It enforces the protocol's ordering rules instead of trusting them: a send buffer may only be rewritten after the
    collective that reads it has been waited for, a gathered buffer only read after its collective has been waited for."""

    def __init__(self, every):
        self.every = every
        self.Y_loc = [np.full((every * mp, g.dim), np.nan) for _ in range(2)]
        self.Y_all = [np.full((world * every * mp, g.dim), np.nan) for _ in range(2)]
        self.work = [None, None]
        self.in_flight = [False, False]
        self.parts = [0, 0]
        self.steps_of_slot = [[], []]
        self.expanded = []
        self.step = 0

    def params(self, step):
        return a0 * (1.0 + step)       # every step sweeps different parameters: a stale buffer cannot pass

    def wait_slot(self, k):
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        self.in_flight[k] = False

    def solve_local(self, k, j):
        assert not self.in_flight[k], f"slot {k} rewritten while its all-gather is in flight"
        if j == 0:
            self.steps_of_slot[k] = []
        part = self.Y_loc[k][j * mp:(j + 1) * mp]
        part[:] = 0.0                  # a short shard is padded with zero vectors
        if hi > lo:
            part[:hi - lo] = ro.generate_solutions(g, self.params(self.step)[lo:hi])
        self.steps_of_slot[k].append(self.step)

    def allgather_async(self, k, n):
        assert not self.in_flight[k] and n == len(self.steps_of_slot[k])
        self.work[k] = dist.all_gather_into_tensor(torch.from_numpy(self.Y_all[k][:world * n * mp]),
                                                   torch.from_numpy(self.Y_loc[k][:n * mp]), async_op=True)
        self.in_flight[k] = True
        self.parts[k] = n

    def expand(self, k, j):
        self.expanded.append((self.step, self.Y_loc[k][j * mp:j * mp + hi - lo].copy()))   # rows of the own shard
        self.step += 1

    def drain(self):
        for k in range(2):
            self.wait_slot(k)

    def check_gathered(self, k):
        """slot k's last collective, once waited for: rank r, part j = rows [r mp, ...) of the unsharded sweep of that step"""
        self.wait_slot(k)
        n = self.parts[k]
        for jpart, st in enumerate(self.steps_of_slot[k][:n]):
            ref = ro.generate_solutions(g, self.params(st))
            for r in range(world):
                rlo, rhi = sweep.shard_bounds(M, world, r)
                got = self.Y_all[k][(r * n + jpart) * mp:(r * n + jpart + 1) * mp]
                assert np.array_equal(got[:rhi - rlo], ref[rlo:rhi]), f"step {st}: gathered shard of rank {r} differs"
                assert not got[rhi - rlo:].any(), "padding rows of a short shard must be zero"


for every in (1, 3):
    be = GlooBackend(every)
    steps = 7
    for s in range(steps):
        k = sweep.run_step(be, s, every)
        assert k == (s // every) % 2
        if s % every == every - 1 and s >= every:
            be.check_gathered(k ^ 1)   # the PREVIOUS group's gathered block (other slot): complete and equal to the sweep
    sweep.drain(be, steps, every)      # (7 steps in groups of 3: the last group has one shard)
    kl = ((steps - 1) // every) % 2
    assert be.parts[kl] == (steps % every or every)
    be.check_gathered(kl)
    assert [st for st, _ in be.expanded] == list(range(steps))
    for st, rows in be.expanded:
        assert np.array_equal(rows, ro.generate_solutions(g, be.params(st))[lo:hi])
    # the protocol check itself must bite: rewriting a slot without waiting is refused
    be.steps_of_slot[0] = [0]
    be.allgather_async(0, 1)
    try:
        be.solve_local(0, 0)
        raise SystemExit("missing wait was not detected")
    except AssertionError:
        pass
    sweep.drain(be)
    dist.barrier()
dist.destroy_process_group()
print("OK", rank)
'''


@pytest.mark.parametrize("M", [8, 7])
def test_step_loop_two_ranks_gloo(tmp_path, M):
    """The double-buffered step loop bench.py runs for N > 1 (sweep.run_step: wait-before-rewrite, solve, asynchronous
    all-gather once per group of steps, expansion, slot alternation) driven by two processes through a gloo stand-in for
    the RCCL calls: over 7 steps with different parameters per step, exchanged every step and in groups of 3 (the last
    group incomplete), for an even and a ragged M, the gathered shards of every step equal the unsharded sweep (the
    reference's map over all parameters, src/lib/SolutionsManagers.py:51,64-68)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "step_worker.py"
    script.write_text(_STEP_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r), str(M)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "OK" in o


def test_gpu_step_backend_follows_the_protocol():
    """GpuStepBackend against a recording fake of the ctypes layer: the call sequence of run_step on the real backend
    class (no GPU needed) -- wait(slot) before the solve that rewrites the slot, gather after the solve, expansion last;
    a short shard solves / expands only its valid rows and zero-fills its padded send buffers once."""
    from romhighcontrast_amd import sweep
    log = []

    class Buf:
        def __init__(self, n):
            self.n = n

        def fill(self, v):
            log.append(("fill", self.n, v))

    class Ctx:
        def alloc(self, n):
            return Buf(n)

        def comm_wait_slot(self, k):
            log.append(("wait_slot", k))

        def allgather_async(self, send, so, recv, ro_, count, slot=0):
            log.append(("allgather", send.n, recv.n, count, slot))

        def comm_wait(self, host):
            log.append(("comm_wait", host))

        def solve_status(self):
            log.append(("status",))

    class Fem:
        reduced_stride, compact_stride, dim = 10, 6, 100

        def solve_reduced(self, a, M, Y, y_row0=0):
            log.append(("solve", M, Y.n, y_row0))

        def allgather_packed_async(self, Y, M, send, recv, recv_off, slot=0):
            log.append(("pack+allgather", M, Y.n, send.n, recv.n, slot))

        def expand(self, a, M, Y, U, y_row0=0):
            log.append(("expand", M, Y.n, U.n, y_row0))

    be = sweep.GpuStepBackend(Ctx(), Fem(), "a", 4, 2, m_valid=3)
    assert log == [("fill", 40, 0.0), ("fill", 40, 0.0)]
    del log[:]
    for s in range(3):
        assert sweep.run_step(be, s) == s % 2
    sweep.drain(be, 3, 1)
    # the solve writes full-stride vectors (4 x 10), what is gathered is their compact form (4 x 6 -> 2 ranks x 4 x 6), all
    # M rows of the shard (the padding row of the short shard included); the own rows are expanded from the full vectors
    per_step = [("wait_slot", None), ("solve", 3, 40, 0), ("pack+allgather", 4, 40, 24, 48, None), ("expand", 3, 40, 400, 0)]
    want = []
    for s in range(3):
        for rec in per_step:
            want.append(tuple(s % 2 if v is None else v for v in rec))
    assert log == want + [("status",), ("comm_wait", True)], log
    # groups of 3 steps: one wait and one collective per group, the parts of a group side by side in the slot's buffers
    # (3 x 4 x 10 full vectors, 3 x 4 x 6 packed, 2 ranks x 3 x 4 x 6 gathered); 4 steps: the last group sends one part
    del log[:]
    be = sweep.GpuStepBackend(Ctx(), Fem(), "a", 4, 2, every=3)
    for s in range(4):
        assert sweep.run_step(be, s, 3) == (s // 3) % 2
    sweep.drain(be, 4, 3)
    assert log == [("wait_slot", 0), ("solve", 4, 120, 0), ("expand", 4, 120, 400, 0),
                   ("solve", 4, 120, 4), ("expand", 4, 120, 400, 4),
                   ("solve", 4, 120, 8), ("pack+allgather", 12, 120, 72, 144, 0), ("expand", 4, 120, 400, 8),
                   ("wait_slot", 1), ("solve", 4, 120, 0), ("expand", 4, 120, 400, 0),
                   ("pack+allgather", 4, 120, 72, 144, 1), ("status",), ("comm_wait", True)], log
    assert be.parts == [3, 1]


def test_parameter_sampler_matches_reference():
    """get_a2test_and_train's sampling (src/experiments/HighContrast.py:99-115) against the reference's
    own output for three (geometry, groups, seed) settings (fixture g8)."""
    from conftest import load_golden
    from romhighcontrast_amd.experiments import get_full_a, sample_parameters
    z = load_golden("g8_experiment.npz")
    a, ahc = sample_parameters((2, 2), [[(0, 0), (1, 1)], [(0, 1)]], 2, 30, 7)
    assert np.array_equal(a, z["a"]) and np.array_equal(ahc, z["a_high_contrast"])
    a, ahc = sample_parameters((3, 3), [[(1, 1)]], 3, 25, 42)
    assert np.array_equal(a, z["s1_a"]) and np.array_equal(ahc, z["s1_ahc"])
    a, ahc = sample_parameters((2, 2), [[(0, 0)], [(1, 1)], [(0, 1), (1, 0)]], 1, 20, 3)
    assert np.array_equal(a, z["s2_a"]) and np.array_equal(ahc, z["s2_ahc"])
    assert np.array_equal(get_full_a(np.array([[2.0, 3.0]]), (2, 2), [[(0, 0)], [(1, 1)]]),
                          np.array([[[2.0, 1.0], [1.0, 3.0]]]))


def test_bench_cpu_baselines_run_on_a_small_case():
    """bench.py's two CPU figures (oracle path on one core / on a process pool) on a tiny geometry:
    they must run without a GPU, report what they used, and the pool must not lose or duplicate parameters."""
    import bench
    blocks, N, M = (2, 2), 8, 24
    a = 10.0 ** np.random.default_rng(0).uniform(0, 2, size=(M,) + blocks)
    one = bench.cpu_baseline(blocks, N, a, budget_s=0.2)
    assert one["cores"] == 1 and one["kind"] == "port" and one["value"] > 0 and one["unit"] == "solves/s"
    pool = bench.cpu_baseline_pool(blocks, N, a, per_worker=2)
    assert pool["kind"] == "port" and pool["value"] > 0 and 1 <= pool["cores"] <= 16
    n_done = int(pool["sample"].split(" of the ")[0])
    assert n_done == min(M, pool["cores"] * 2)


_STUB_RANK = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                      "ROMHC_LAUNCH_ID", "HSA_ENABLE_IPC_MODE_LEGACY")}
rec["argv"] = sys.argv[1:]
open(os.path.join(os.environ["STUB_DIR"], f"rank{rank}.json"), "w").write(json.dumps(rec))
if os.environ.get("STUB_FAIL_RANK") == str(rank):
    sys.exit(7)
if os.environ.get("STUB_FAIL_RANK") is not None:
    time.sleep(60)   # the surviving ranks would hang in a collective: the launcher must stop them
print(json.dumps({"n_gpus": world}) if rank == 0 else f"noise from rank {rank}")
'''


def test_bench_launcher_starts_one_rank_per_gpu(tmp_path, monkeypatch):
    """`python bench.py --gpus N` without a launcher environment starts N fresh ranks itself (before anything touches
    the GPU): distinct RANK / LOCAL_RANK, one MASTER_PORT and launch id, only rank 0 on stdout, the first failing rank's
    exit code returned and the other ranks stopped."""
    import json
    import time
    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB_RANK)
    env = dict(os.environ, STUB_DIR=str(tmp_path), PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    code = ("import sys, bench; sys.exit(bench.launch_ranks(3, ['--gpus', '3', '--steps', '2'], "
            f"child=[sys.executable, {str(stub)!r}]))")
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip()) == {"n_gpus": 3}           # ONE line on stdout: rank 0's
    assert "noise from rank 1" in p.stderr and "noise from rank 2" in p.stderr
    recs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and {r["LOCAL_WORLD_SIZE"] for r in recs} == {"3"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and len({r["ROMHC_LAUNCH_ID"] for r in recs}) == 1
    assert {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"} and {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in recs} == {"0"}
    assert recs[0]["argv"] == ["--gpus", "3", "--steps", "2"]
    # the launcher itself never loads libromhc (no GPU initialisation in the parent)
    code2 = ("import sys, bench; rc = bench.launch_ranks(2, [], child=[sys.executable, " + repr(str(stub)) + "]); "
             "assert not any('libromhc' in l for l in open('/proc/self/maps')); sys.exit(rc)")
    assert subprocess.run([sys.executable, "-c", code2], env=env, cwd=ROOT, capture_output=True, timeout=120).returncode == 0
    # a failing rank: its exit code comes back and the sleeping ranks are stopped long before their 60 s
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], env=dict(env, STUB_FAIL_RANK="1"), cwd=ROOT, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 7 and time.time() - t0 < 30
    assert "rank 1" in p.stderr


def test_bench_refuses_more_ranks_than_gpus(monkeypatch):
    import bench
    for k in ("ROMHC_FORCE_DEVICE", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROMHC_LAUNCH_ID"):
        monkeypatch.delenv(k, raising=False)
    assert bench.pick_device(3, 8, 8) == 3
    with pytest.raises(SystemExit, match="one process per GPU"):
        bench.pick_device(1, 2, 1)                     # two ranks, one visible GPU: no modulo
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5")     # a launcher that pinned one GPU per process
    assert bench.pick_device(5, 8, 1) == 0
    monkeypatch.setenv("ROMHC_LAUNCH_ID", "x")         # bench.py's own launcher never pins: the job HAS one GPU
    with pytest.raises(SystemExit, match="one process per GPU"):
        bench.pick_device(1, 2, 1)
    monkeypatch.setenv("ROMHC_FORCE_DEVICE", "0")
    assert bench.pick_device(1, 2, 1) == 0


def test_bench_workloads_follow_survey_8d():
    import bench
    a = bench.workload_parameters("c2", (2, 2), 1024)
    assert np.array_equal(a, 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(1024, 2, 2)))
    a8 = bench.workload_parameters("c2", (2, 2), 8192)   # C3: rank r owns rows [1024 r, 1024 (r+1))
    assert a8.shape == (8192, 2, 2) and a8.min() >= 1 and a8.max() <= 100
    c4 = bench.workload_parameters("c4", (3, 3), 1024)
    assert np.all(c4[0] == 1) and all(c4[1 + j].flat[j] == 1e8 and (c4[1 + j] == 1).sum() == 8 for j in range(9))
    assert np.all(c4[10] == 1e8) and c4[11:].min() >= 1 and c4[11:].max() <= 1e8
    assert np.array_equal(c4[11:], 10.0 ** np.random.default_rng(20240807).uniform(0, 8, size=(1013, 3, 3)))
    c5 = bench.workload_parameters("c5", (4, 4), 4096)
    assert c5.shape == (4096, 4, 4) and c5.max() <= 1000


def test_environment_switches_are_documented():
    """Every ROMHC_* variable the library reads (getenv in csrc/, os.environ in the Python layer) has a row in
    INTEGRATION.md, and the table names no variable that nothing reads."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "romhighcontrast_amd", "csrc")
    read = set()
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            read |= set(re.findall(r'getenv\("(ROMHC_[A-Z0-9_]+)"\)', open(os.path.join(csrc, fn)).read()))
    for rel in ("romhighcontrast_amd/_ffi.py", "romhighcontrast_amd/lib/ReducedBasis.py", "romhighcontrast_amd/sweep.py"):
        read |= set(re.findall(r'environ(?:\.get)?[\[(]\s*"(ROMHC_[A-Z0-9_]+)"', open(os.path.join(root, rel)).read()))
    doc = set(re.findall(r"ROMHC_[A-Z0-9_]+", open(os.path.join(root, "INTEGRATION.md")).read()))
    dev_only = {"ROMHC_EXT_LDS_PAD"}                      # timeline builds (-DROMHC_STAMPS): tools/README.md
    # (the forcing switches of the A/B build go through ab_env(...), not getenv("..."): the product library reads none of them)
    assert not re.findall(r'getenv\("ROMHC_(?:NO_EXT_LR|NO_FOLD_EXPAND|EXT_FLAT|X128_SYS_FAST|NO_TILE_PAIRS|NO_TILE_STREAM)"\)',
                          "".join(open(os.path.join(csrc, fn)).read() for fn in os.listdir(csrc) if fn.endswith((".hip", ".h"))))
    launcher = {"ROMHC_LAUNCH_ID", "ROMHC_FORCE_DEVICE"}  # set / read by bench.py's launcher, not by the library
    ab_build = set(re.findall(r'ab_env\("(ROMHC_[A-Z0-9_]+)"\)', open(os.path.join(csrc, "rom_fem_setup.hip")).read())) | {"ROMHC_AB"}
    assert ab_build == {"ROMHC_AB", "ROMHC_NO_EXT_LR", "ROMHC_NO_FOLD_EXPAND", "ROMHC_EXT_FLAT", "ROMHC_X128_SYS_FAST", "ROMHC_NO_TILE_PAIRS",
                        "ROMHC_NO_TILE_STREAM", "ROMHC_COEF_GLOBAL"}, ab_build   # (named in INTEGRATION.md as retired: read by libromhc_ab.so only)
    launcher |= ab_build
    assert read - dev_only <= doc, sorted(read - dev_only - doc)
    assert doc - launcher <= read, sorted(doc - launcher - read)


def _build_c_caller(tmp_path):
    """gcc -std=c99 build of tests/c_abi/c_abi_smoke.c against include/romhc.h and the in-tree libromhc.so."""
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.join(root, "romhighcontrast_amd", "csrc")
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
           os.path.join(root, "tests", "c_abi", "c_abi_smoke.c"), "-L", libdir, "-lromhc", "-lm", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    return exe, dict(os.environ, LD_LIBRARY_PATH=libdir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))


def test_plain_c_caller_builds_against_the_header(tmp_path):
    """include/romhc.h is a C header (no C++, no torch types): a C99 program that calls the sweep, the norms and the
    single-call basis stage compiles with -Wall -Wextra -Werror, links against libromhc.so and starts; without a GPU it
    is told so through rom_last_error() (the compute part of the same program is a -m gpu test)."""
    exe, env = _build_c_caller(tmp_path)
    r = subprocess.run([exe, "--symbols"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode()
    assert r.returncode == 0 and "libromhc version" in out and "rom_device_count ->" in out, out



def test_host_array_registry_is_by_identity_and_verified():
    """`generate_solutions` remembers the interface vectors of the (two most recent) host arrays it returned; a builder handed
    such an array back gets them only if it IS that array (not a copy) and the uploaded rows still equal their expansion bit
    for bit.  The bookkeeping is host logic: exercised here with stand-ins for the device objects."""
    import gc
    import pickle
    from romhighcontrast_amd.lib import SolutionsManagers as SM

    class Buf:
        def __init__(self, same):
            self.same, self.asked = same, 0

        def same_bits_as(self, other, n):
            self.asked += 1
            return self.same

    class Rows:
        def __init__(self, same):
            self.buf = Buf(same)

    class Interface:
        def __init__(self, M, same=True):
            self.M, self.same, self.expanded = M, same, 0

        def rows(self):
            self.expanded += 1
            return Rows(self.same)

    class Uploaded:
        def __init__(self, rows, dim):
            self.rows, self.dim, self.buf = rows, dim, object()

    sm = object.__new__(SM.SolutionsManagerFEM)
    sm.vspace_dim = 9
    a1, a2, a3 = np.zeros((4, 9)), np.zeros((4, 9)), np.zeros((5, 9))
    f1, f2, f3 = Interface(4), Interface(4, same=False), Interface(5)
    assert sm.factored_of_host_rows(a1, Uploaded(4, 9)) is None          # nothing remembered yet
    sm._remember_host_block(a1, f1)
    assert sm.factored_of_host_rows(a1, Uploaded(4, 9)) is f1 and f1.expanded == 1
    assert sm.factored_of_host_rows(a1.copy(), Uploaded(4, 9)) is None   # a copy is not the array
    assert sm.factored_of_host_rows(a1[:3], Uploaded(3, 9)) is None      # nor is a slice
    assert sm.factored_of_host_rows(a1, Uploaded(4, 8)) is None          # another space
    sm._remember_host_block(a2, f2)
    assert sm.factored_of_host_rows(a2, Uploaded(4, 9)) is None and f2.expanded == 1   # rows were written into: verified, refused
    assert sm.factored_of_host_rows(a1, Uploaded(4, 9)) is f1            # the one before is still there
    sm._remember_host_block(a3, f3)
    assert sm.factored_of_host_rows(a1, Uploaded(4, 9)) is None          # two most recent only
    assert sm.factored_of_host_rows(a3, Uploaded(5, 9)) is f3
    del a2
    gc.collect()
    sm._remember_host_block(a1, f1)                                      # dead arrays leave the list
    assert all(ref() is not None for ref, _ in sm._host_blocks) and len(sm._host_blocks) <= 2
    assert "_host_blocks" not in sm.__getstate__()                       # device state is not pickled
    pickle.dumps(sm.__getstate__())
