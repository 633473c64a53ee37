"""C5 / C4 sweep time against the number of systems per workspace chunk (does a factor workspace that fits the 256 MiB
Infinity Cache pay?).  usage: gpu_chunk_sweep.py [c5|c4]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from romhighcontrast_amd.lib import SolutionsManagers as SM

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
blocks, N, M = {"c5": ((4, 4), 256, 4096), "c4": ((3, 3), 171, 1024)}[cfg]
sm = SM.SolutionsManagerFEM(blocks, N)
ctx, fem, dim = sm._ctx, sm._fem, sm.vspace_dim
a = bench.workload_parameters(cfg, blocks, M)
a_dev = ctx.upload(np.ascontiguousarray(a))
U = ctx.alloc(M * dim)


for mb in (24 << 10, 4096, 2048, 1024, 512, 256, 192, 128, 96, 64):
    ctx.set_workspace_limit(mb << 20)
    for rep in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        fem.solve_batch(a_dev, M, U, 0, wait=False)
        ctx.synchronize()
        w = time.perf_counter() - t0
    print(f"limit {mb:6d} MB: {w*1e3:8.3f} ms  {M/w:10.0f} solves/s", flush=True)
