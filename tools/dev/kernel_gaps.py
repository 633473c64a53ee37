"""Idle time between the kernels of a sweep (dev tool): reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv) and prints, per
kernel name, its busy time and the gap in front of it (previous kernel's end -> this kernel's start) over the steady part.
usage: python tools/dev/kernel_gaps.py path/to/*_kernel_trace.csv [skip_first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
busy, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = None
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0][:40]
    busy[n] += e - s
    cnt[n] += 1
    if prev_end is not None:
        gap[n] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
tot = t1 - t0
print(f"span {tot * 1e-6:.3f} ms, busy {sum(busy.values()) * 1e-6:.3f} ms (overlapping kernels counted twice), gaps {sum(gap.values()) * 1e-6:.3f} ms = {100 * sum(gap.values()) / tot:.1f} %")
for n, b in busy.most_common():
    print(f"  {n:42s} {cnt[n]:6d} x {b / cnt[n] * 1e-3:8.1f} us   gap in front {gap[n] / cnt[n] * 1e-3:6.2f} us")
