"""Phase timing inside the extension kernel from in-kernel cycle stamps, with per-CU timelines (dev tool; needs a library
built with `make -C romhighcontrast_amd/csrc EXTRA=-DROMHC_STAMPS`).  env: NB, N, M as tools/gpu_ext_time.py; the
kernel variant is chosen by the usual switches (ROMHC_EXT_W3, ...)."""
import os
import sys
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romhighcontrast_amd import _ffi

ctx = _ffi.get_context(0)
lib = ctx.lib
NB, N, M = int(os.environ.get("NB", "2")), int(os.environ.get("N", "128")), int(os.environ.get("M", "1024"))
fem = _ffi.Fem(ctx, NB, NB, N)
a = 10.0 ** np.random.default_rng(20240807).uniform(0, 2, size=(M, NB * NB))
ab, U = ctx.upload(a), ctx.alloc(M * fem.dim)
Y = ctx.alloc(M * fem.reduced_stride)
fem.solve_reduced(ab, M, Y)
for _ in range(3):
    fem.expand(ab, M, Y, U)
ctx.synchronize()
lib.rom_debug_stamps_clear()
ctx.timer_start()
fem.expand(ab, M, Y, U)
ms = ctx.timer_stop()
n = 16384 * 6
buf = (C.c_ulonglong * n)()
assert lib.rom_debug_stamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 6).astype(np.int64)
t = t[(t[:, 4] > t[:, 0]) & (t[:, 0] > 0)]
hw = t[:, 5]
cu = ((hw >> 16) & 0xf) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)  # xcc, se, sh, cu
waitcol = os.environ.get("WAITCOL")  # library built with -DROMHC_STAMPS_WAIT: column 1 = cycles in the k loop's waits
if waitcol:
    wait = t[:, 1].copy()
    t[:, 1] = t[:, 0]
d = np.diff(t[:, :5], axis=1)
names = ["entry -> first loads issued", "-> first barrier passed", "-> k loop done", "-> stores issued (end)"]
life = t[:, 4] - t[:, 0]
print(f"{len(t)} workgroups on {len(np.unique(cu))} CUs, launch {ms * 1e3:.1f} us; lifetime mean {life.mean():.0f} cycles, median {np.median(life):.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:32s} mean {d[:, i].mean():8.0f}  median {np.median(d[:, i]):8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
if waitcol:
    kl = t[:, 3] - t[:, 2]
    print(f"  k loop {kl.mean():.0f} cycles, of which wave 0 waits for loads + barrier: mean {wait.mean():.0f}  median {np.median(wait):.0f}  "
          f"p10 {np.percentile(wait, 10):.0f}  p90 {np.percentile(wait, 90):.0f}  ({wait.sum() / kl.sum():.2f} of the k loop)")
# per-CU timelines: how much of a CU's span has 0 / 1 / >= 2 workgroups inside the k loop, and the gap between a
# workgroup's last stamp and the entry of the workgroup that takes its place
kcover = np.zeros(4)
spans, gaps, conc = [], [], []
for c in np.unique(cu):
    w = t[cu == c]
    w = w[np.argsort(w[:, 0])]
    t0, t1 = w[:, 0].min(), w[:, 4].max()
    spans.append(t1 - t0)
    ev = sorted([(x, +1) for x in w[:, 2]] + [(x, -1) for x in w[:, 3]])
    level, last = 0, t0
    for x, s in ev:
        kcover[min(level, 3)] += x - last
        level, last = level + s, x
    kcover[0] += t1 - last
    # resident workgroups over time -> replacement gaps
    ev = sorted([(x, +1) for x in w[:, 0]] + [(x, -1) for x in w[:, 4]])
    level, peak = 0, 0
    ends = []
    for x, s in ev:
        if s < 0:
            ends.append(x)
        elif ends:
            gaps.append(x - ends.pop(0))
        level += s
        peak = max(peak, level)
    conc.append(peak)
spans = np.array(spans)
gaps = np.array(gaps)
print(f"per-CU span mean {spans.mean():.0f} cycles (max {spans.max()}); workgroups resident at once per CU: max {max(conc)}")
print("share of the CUs' spans with 0 / 1 / 2 / >=3 workgroups inside the k loop: " + " / ".join(f"{x / kcover.sum():.3f}" for x in kcover))
if len(gaps):
    print(f"last stamp of a workgroup -> entry of its successor on the CU: mean {gaps.mean():.0f} cycles, median {np.median(gaps):.0f}, "
          f"p10 {np.percentile(gaps, 10):.0f}, p90 {np.percentile(gaps, 90):.0f}")
c0 = np.unique(cu)[3]
w = t[cu == c0]
w = w[np.argsort(w[:, 0])][:14]
print("one CU, first workgroups (cycles from the CU's first entry): entry / k loop start / k loop end / stores issued")
for r in w:
    print("   " + "  ".join(f"{x - w[0, 0]:8d}" for x in (r[0], r[2], r[3], r[4])))
