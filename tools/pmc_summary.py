"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into profiles/*.json.

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide
(16 B per lane) coalesced reads, so it is doubled (MI355X_MICROARCH.md, section HBM).  Infinity-Cache hits
are included in both counters, so "traffic" is memory-side (fabric) traffic, an upper bound on HBM bytes.
usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16  # noqa: E402  (hash of the kernel sources: bench.py withholds `traffic` taken on another build)


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return d


f, w = agg(sys.argv[1]), agg(sys.argv[2])
out = {}
for k in f:
    if k.startswith("__amd"):
        continue
    fa = sum(f[k]) / len(f[k]) * 1024.0
    wa = sum(w.get(k, [0.0])) / max(1, len(w.get(k, [0.0]))) * 1024.0
    out[k] = {"launches_sampled": len(f[k]), "fetch_bytes_raw": fa, "fetch_bytes_x2": 2 * fa, "write_bytes": wa,
              "traffic_bytes_per_launch": 2 * fa + wa}
json.dump({"unit": "bytes per launch (average over the sampled launches)", "csrc_sha16": csrc_sha16(),
           "correction": "FETCH_SIZE x 2 (gfx950, 16-B-per-lane reads); WRITE_SIZE as reported; MALL hits included",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
