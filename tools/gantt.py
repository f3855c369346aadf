"""Coarse Gantt chart of the multi-stream step from a rocprofv3 --kernel-trace CSV: the last `nsteps` steps cut into bins; per bin the
number of kernels in flight (time-averaged), the share of the bin with >= 1 chip-filling kernel (>= 200 workgroups) in flight, and
the kernels that occupy the bin, by queue.  usage: gantt.py <kernel_trace.csv> [bin_us=250] [span_ms=34]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
bin_ns = int(float(sys.argv[2]) * 1e3) if len(sys.argv) > 2 else 250000
span_ns = int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else 34000000
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = 1
    for ax in "XYZ":
        g, w = int(r.get(f"Grid_Size_{ax}", 1) or 1), int(r.get(f"Workgroup_Size_{ax}", 1) or 1)
        wg *= max(1, g // max(1, w))
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    ev.append((s, e, name[:34], wg, r.get("Queue_Id", "0")))
ev.sort()
t_end = ev[-1][1]
t0 = t_end - span_ns
ev = [x for x in ev if x[1] > t0]
nb = span_ns // bin_ns
print(f"bins of {bin_ns/1e3:.0f} us over the last {span_ns/1e6:.0f} ms; columns: t(ms) | avg kernels in flight | share with a >=200-WG kernel | busy share | kernels (queue:name xshare)")
for b in range(nb):
    lo, hi = t0 + b * bin_ns, t0 + (b + 1) * bin_ns
    occ = collections.defaultdict(float)
    pts = []
    big = []
    for s, e, n, wg, q in ev:
        a, z = max(s, lo), min(e, hi)
        if z > a:
            occ[(q, n, wg >= 200)] += z - a
            pts.append((a, 1)); pts.append((z, -1))
            if wg >= 200:
                big.append((a, z))
    tot = sum(occ.values())
    # union helpers
    def union(iv):
        iv.sort(); u = 0; cs = ce = None
        for a, z in iv:
            if cs is None: cs, ce = a, z
            elif a > ce: u += ce - cs; cs, ce = a, z
            else: ce = max(ce, z)
        return u + (ce - cs if cs is not None else 0)
    allu = union([(max(s, lo), min(e, hi)) for s, e, *_ in ev if min(e, hi) > max(s, lo)])
    bigu = union(big)
    top = sorted(occ.items(), key=lambda kv: -kv[1])[:5]
    desc = "  ".join(f"{q}:{n}{'*' if isbig else ''} {v / bin_ns:.2f}" for (q, n, isbig), v in top)
    print(f"{b * bin_ns / 1e6:6.2f} | {tot / bin_ns:4.2f} | {bigu / bin_ns:4.2f} | {allu / bin_ns:4.2f} | {desc}")
