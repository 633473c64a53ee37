"""Where the N > 1 step spends its extra time against the fused single-GPU step (one rank, C2): fused sweep / two-stage
sweep / + pack / + RCCL all-gather (1-rank communicator), ms per step over 200 steps.  env: NCCL_* pass through to RCCL."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from romhighcontrast_amd import _ffi, sweep

ctx = _ffi.get_context()
blocks, N, M = (2, 2), 128, 1024
fem = _ffi.Fem(ctx, blocks[0], blocks[1], N)
a = bench.workload_parameters("c2", blocks, M)
a_dev = ctx.upload(np.ascontiguousarray(a))
U = ctx.alloc(M * fem.dim)
Y = [ctx.alloc(M * fem.reduced_stride) for _ in range(2)]
Yc = [ctx.alloc(M * fem.compact_stride) for _ in range(2)]
Yc_all = [ctx.alloc(M * fem.compact_stride) for _ in range(2)]
ctx.comm_init(ctx.comm_unique_id(), 0, 1)
K = 200
G = int(os.environ.get("EVERY", "4"))
Yg = [ctx.alloc(G * M * fem.reduced_stride) for _ in range(2)]
Ycg = [ctx.alloc(G * M * fem.compact_stride) for _ in range(2)]
Ycg_all = [ctx.alloc(G * M * fem.compact_stride) for _ in range(2)]
def run_grouped():
    ctx.synchronize(); ctx.comm_wait(True)
    t0 = time.perf_counter()
    for s in range(K):
        g, j = divmod(s, G)
        k = g & 1
        if j == 0:
            ctx.comm_wait_slot(k)
        fem.solve_reduced(a_dev, M, Yg[k], y_row0=j * M)
        if j == G - 1:
            fem.allgather_packed_async(Yg[k], G * M, Ycg[k], Ycg_all[k], 0, slot=k)
        fem.expand(a_dev, M, Yg[k], U, y_row0=j * M)
    t_enq = (time.perf_counter() - t0) / K * 1e3
    ctx.solve_status(); ctx.comm_wait(True)
    return (time.perf_counter() - t0) / K * 1e3, t_enq

def run(mode):
    ctx.synchronize(); ctx.comm_wait(True)
    t0 = time.perf_counter()
    for s in range(K):
        k = s & 1
        if mode == "fused":
            fem.solve_batch(a_dev, M, U, wait=False)
            continue
        if mode in ("gather", "packed"):
            ctx.comm_wait_slot(k)
        fem.solve_reduced(a_dev, M, Y[k])
        if mode in ("pack", "gather"):
            fem.pack_reduced(Y[k], M, Yc[k])
        if mode == "gather":
            ctx.allgather_async(Yc[k], 0, Yc_all[k], 0, M * fem.compact_stride, slot=k)
        if mode in ("packed", "packed2"):
            fem.allgather_packed_async(Y[k], M, Yc[k], Yc_all[k], 0, slot=k)
        if mode == "packed2":   # the wait for the OTHER slot right behind the record (adjacent packets in the compute queue)
            ctx.comm_wait_slot(k ^ 1)
        fem.expand(a_dev, M, Y[k], U)
    t_enq = (time.perf_counter() - t0) / K * 1e3
    ctx.solve_status(); ctx.comm_wait(True)
    return (time.perf_counter() - t0) / K * 1e3, t_enq
for mode in ("fused", "packed", "packed2", "fused", "packed", "packed2"):
    t, te = run(mode)
    print(f"{mode:10s} {t:.4f} ms per step   (host enqueue {te:.4f} ms per step)", flush=True)
t, te = run_grouped()
print(f"grouped x{G} {t:.4f} ms per step   (host enqueue {te:.4f} ms per step)", flush=True)
t, te = run_grouped()
print(f"grouped x{G} {t:.4f} ms per step   (host enqueue {te:.4f} ms per step)", flush=True)
ctx.comm_destroy()
