"""Launch-order timeline of the LAST call in a rocprofv3 --kernel-trace CSV (dev tool): the kernels from the last launch whose
name contains MARK to the end (or to the next launch containing END), with start offset, duration and the idle gap in front.
usage: python tools/dev/kernel_timeline.py kernel_trace.csv MARK [END]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2]
end = sys.argv[3] if len(sys.argv) > 3 else None
starts = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
i0 = starts[-1]
i1 = len(rows)
if end:
    for i in range(i0 + 1, len(rows)):
        if end in rows[i]["Kernel_Name"]:
            i1 = i
            break
t0 = int(rows[i0]["Start_Timestamp"])
prev = None
busy = gaps = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = 0 if prev is None else max(0, s - prev)
    busy += e - s
    gaps += g
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{(s - t0) * 1e-3:9.1f} us  +{g * 1e-3:6.1f}  {(e - s) * 1e-3:8.1f} us  {n}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
    prev = max(prev or 0, e)
print(f"launches {i1 - i0}, span {(prev - t0) * 1e-6:.3f} ms, busy {busy * 1e-6:.3f} ms, gaps {gaps * 1e-6:.3f} ms")
