"""Average SQ / GRBM counters per dispatch and kernel from a rocprofv3 --pmc counter_collection.csv (dev tool).
usage: python tools/pmc_sq_summary.py <counter_collection.csv> [kernel-name substrings ...]"""
import collections
import csv
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2:]
for k in sorted(d):
    if k.startswith("__amd") or (want and not any(w in k for w in want)):
        continue
    n = max(len(v) for v in d[k].values())
    print(f"{k}  ({n} dispatches)")
    for c in sorted(d[k]):
        print(f"  {c:32s} {sum(d[k][c]) / len(d[k][c]):16.0f}")
