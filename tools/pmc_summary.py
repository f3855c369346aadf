#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (…_counter_collection.csv files under a directory): per-kernel launch counts and
per-launch counter averages -> <out>.txt, plus the roofline.traffic json for conv_gemm_bf16_glds_kernel -> <out>.json.
usage: pmc_summary.py <rocprof output dir> <out prefix> "<command line that was profiled>" """
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd.build import source_hash       # the summary names the sources its counters were collected from (bench.py marks others stale)

d, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
per = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = {c.lower(): c for c in rd.fieldnames}
        kn, cn, cv = cols["kernel_name"], cols["counter_name"], cols["counter_value"]
        did = cols.get("dispatch_id") or cols.get("correlation_id")
        for row in rd:
            k = row[kn]
            per[k][row[cn]] += float(row[cv])
            launches[k].add(row[did])
lines = []
for k in sorted(per, key=lambda k: -per[k].get("TCC_EA0_RDREQ_sum", 0.0)):
    n = max(1, len(launches[k]))
    lines.append(f"{k[:56]} launches {n} " + str({c: int(v / n) for c, v in sorted(per[k].items())}))
with open(out + ".txt", "w") as fh:
    fh.write("\n".join(lines) + "\n")
js_all = {}
for k in per:
    sym = k.split("(")[0]
    sym = sym.replace("void ", "")
    if sym.startswith("conv_gemm_bf16_glds") or sym.startswith("conv_gemm_bf16_s64") or "ln_dwconv7" in sym or "dwconv7_ln_fwd" in sym or "wgrad" in sym:
        n = len(launches[k]); c = {a: b / n for a, b in per[k].items()}
        rd_b = c["TCC_EA0_RDREQ_sum"] * 64 * 2          # gfx950: wide streaming reads are counted at half size (guide, HBM section)
        wr_b = c["TCC_EA0_WRREQ_sum"] * 64
        js_all[sym] = {"kernel": sym, "source": os.path.basename(out) + ".txt (" + cmd + "; own pass, no trace domains)",
                       "launches": n, **{a + "_per_launch": int(b) for a, b in c.items()},
                       "read_bytes_per_launch": int(rd_b), "write_bytes_per_launch": int(wr_b),
                       "correction": "MI355X_MICROARCH.md, HBM section: FETCH_SIZE = TCC_EA0_RDREQ x 64 B reports half of a wide (16 B/lane) streaming read on gfx950 -> doubled; WRREQ x 64 B is uncalibrated",
                       "l2_hit_rate": c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
                       "traffic_bytes_per_launch": int(rd_b + wr_b)}
js_all["source_hash"] = source_hash()
with open(out + ".json", "w") as fh:
    json.dump(js_all, fh, indent=1)
print("\n".join(lines[:12]))
