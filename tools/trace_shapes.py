#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid, workgroup, LDS) rows: launches, average / total duration.
    python tools/trace_shapes.py <*_kernel_trace.csv> [min_total_us] > table.txt"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
        g = tuple(int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        w = tuple(int(r[k]) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
        wgs = (g[0] // max(1, w[0])) * (g[1] // max(1, w[1])) * (g[2] // max(1, w[2]))
        key = (name[:70], wgs, w[0], int(r.get("LDS_Block_Size", 0) or 0))
        e = rows[key]
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
print(f"{'kernel':70s} {'WGs':>7s} {'thr':>4s} {'LDS':>7s} {'n':>6s} {'avg us':>9s} {'total us':>10s}")
for (name, wgs, thr, lds), (n, tot) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if tot >= floor:
        print(f"{name:70s} {wgs:7d} {thr:4d} {lds:7d} {n:6d} {tot / n:9.2f} {tot:10.1f}")
