#!/usr/bin/env python3
"""Per-kernel per-launch averages of every counter found in rocprofv3 --pmc output directories.
    python tools/pmc_table.py <dir>[,<dir>...] [kernel-name substring]"""
import collections, csv, glob, os, sys
dirs = sys.argv[1].split(",")
sel = sys.argv[2] if len(sys.argv) > 2 else ""
avg = collections.defaultdict(dict)
for d in dirs:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = {c.lower(): c for c in rd.fieldnames}
            kn, cn, cv = cols["kernel_name"], cols["counter_name"], cols["counter_value"]
            did = cols.get("dispatch_id") or cols.get("correlation_id")
            for row in rd:
                if sel in row[kn]:
                    per[row[kn]][row[cn]] += float(row[cv])
                    launches[row[kn]].add(row[did])
    for k in per:
        n = max(1, len(launches[k]))
        for a, b in per[k].items():
            avg[k][a] = (b / n, n)
for k in sorted(avg):
    print(k.split("(")[0][:90])
    for a in sorted(avg[k]):
        print(f"    {a:36s} {avg[k][a][0]:16.0f}   ({avg[k][a][1]} launches)")
