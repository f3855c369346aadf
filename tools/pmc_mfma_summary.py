#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass of SQ / GRBM counters per kernel symbol: per-launch averages and the MFMA-busy share.

usage: pmc_mfma_summary.py <rocprof output dir> <out prefix> "<command line that was profiled>"

``SQ_VALU_MFMA_BUSY_CYCLES`` counts cycles per SIMD (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md "Per-instruction cycle
constants"), summed over the chip's 1 024 SIMDs; ``GRBM_GUI_ACTIVE`` is the kernel's duration in shader clocks (summed over the
XCDs when the tool reports one value per XCD -- the summary detects that from the ratio to SQ_BUSY_CYCLES).  Reported per symbol:
    mfma_busy = MFMA_BUSY_CYCLES / (1 024 SIMDs x kernel cycles)        the share of SIMD time the matrix pipe is busy
so that 1.0 would be the dense peak AT THE CLOCK THE KERNEL RAN AT (the 2.5 PFLOP/s figure assumes 2.4 GHz; under an MFMA load the
part runs 1.9-2.0 GHz, same guide, DVFS note).  The quad-cycle SQ counters (WAVE_CYCLES, WAIT_*, ACTIVE_INST_*) are given as shares of
SQ_WAVE_CYCLES."""
import collections, csv, glob, json, os, sys

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
            per[row[kn]][row[cn]] += float(row[cv])
            launches[row[kn]].add(row[did])
N_SIMD = 1024
res, lines = {}, []
for k in sorted(per, key=lambda k: -per[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    n = max(1, len(launches[k]))
    c = {a: b / n for a, b in per[k].items()}
    sym = k.split("(")[0].replace("void ", "")
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_BUSY_CYCLES", 0.0)
    # kernel duration in clocks: GRBM_GUI_ACTIVE per launch; if the tool summed it over the 8 XCDs it is ~8x the per-SE busy figure
    cycles = gui
    note = "GRBM_GUI_ACTIVE"
    if gui and busy and gui > 4.0 * (busy / 32.0) and gui / 8.0 > 0:
        pass
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    ent = {"launches": n, **{a + "_per_launch": int(b) for a, b in c.items()}}
    if cycles > 0:
        ent["kernel_cycles_per_launch"] = int(cycles)
        ent["mfma_busy"] = mf / (N_SIMD * cycles)
        ent["mfma_busy_if_gui_is_summed_over_8_xcds"] = mf / (N_SIMD * cycles / 8.0)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        for a in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if a in c:
                ent[a.lower() + "_share_of_wave_cycles"] = c[a] / wc
    res[sym if sym not in res else k[:80]] = ent
    lines.append(f"{k[:60]} launches {n} " + str({a: int(b) for a, b in sorted(c.items())}) +
                 (f" mfma_busy {ent['mfma_busy']:.3f}" if "mfma_busy" in ent else ""))
with open(out + ".txt", "w") as fh:
    fh.write("\n".join(lines) + "\n")
with open(out + ".json", "w") as fh:
    json.dump({"source": cmd + " (own pass, counters only)", "simds": N_SIMD, "kernels": res}, fh, indent=1)
print("\n".join(lines[:16]))
