#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of SQ / GRBM counters per kernel symbol: per-launch averages and the MFMA-busy share.

usage: pmc_mfma_summary.py <rocprof output dir>[,<dir>...] <out prefix> "<command lines that were profiled>"

``SQ_VALU_MFMA_BUSY_CYCLES`` counts cycles per SIMD (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md "Per-instruction cycle
constants"), summed over the chip's 1 024 SIMDs.  ``GRBM_GUI_ACTIVE`` is the kernel's duration in shader clocks, reported summed over
the 8 XCDs (checked on the dominant kernel: 110 GFLOP per launch = 3.36 M MFMAs x 32 = 107 M busy cycles, the counter reads 108 M; a
120 us launch at ~2.2 GHz is 0.27 M clocks, the counter reads 2.16 M = 8 x that).  Reported per symbol:
    mfma_busy = MFMA_BUSY_CYCLES / (1 024 SIMDs x GUI_ACTIVE / 8)        the share of SIMD time the matrix pipe is busy
so that 1.0 would be the dense peak AT THE CLOCK THE KERNEL RAN AT (the 2.5 PFLOP/s figure assumes 2.4 GHz; under an MFMA load the
part clocks lower, same guide, DVFS note).  The quad-cycle SQ counters (WAVE_CYCLES, WAIT_*, ACTIVE_INST_*) are given as shares of
SQ_WAVE_CYCLES: parked (s_waitcnt / barrier), issue-stalled, issuing."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd.build import source_hash       # the summary names the sources its counters were collected from (bench.py marks others stale)

dirs, out, cmd = sys.argv[1].split(","), sys.argv[2], sys.argv[3]
avg = collections.defaultdict(dict)                      # kernel -> counter -> per-launch average
nlaunch = {}
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
                per[row[kn]][row[cn]] += float(row[cv])
                launches[row[kn]].add(row[did])
    for k in per:
        n = max(1, len(launches[k]))
        nlaunch[k] = max(nlaunch.get(k, 0), n)
        for a, b in per[k].items():
            avg[k][a] = b / n
N_SIMD, N_XCD = 1024, 8
res, lines = {}, []
for k in sorted(avg, key=lambda k: -avg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * nlaunch[k]):
    c = avg[k]
    sym = k.split("(")[0].replace("void ", "")
    ent = {"launches": nlaunch[k], **{a + "_per_launch": int(b) for a, b in c.items()}}
    gui, mf = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui > 0:
        ent["kernel_clocks_per_launch"] = int(gui / N_XCD)
        ent["mfma_busy"] = mf / (N_SIMD * gui / N_XCD)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        for a, nm in (("SQ_WAIT_ANY", "waves_parked"), ("SQ_WAIT_INST_ANY", "waves_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "waves_issuing")):
            if a in c:
                ent[nm] = c[a] / wc
    res[sym if sym not in res else k[:80]] = ent
    lines.append(f"{k[:64]} launches {nlaunch[k]} " + (f"mfma_busy {ent['mfma_busy']:.3f} " if "mfma_busy" in ent else "") +
                 " ".join(f"{nm} {ent[nm]:.2f}" for nm in ("waves_parked", "waves_issue_stalled", "waves_issuing") if nm in ent) + " " +
                 str({a: int(b) for a, b in sorted(c.items())}))
with open(out + ".txt", "w") as fh:
    fh.write("\n".join(lines) + "\n")
with open(out + ".json", "w") as fh:
    json.dump({"source_hash": source_hash(), "source": cmd + " (own passes, counters only)", "simds": N_SIMD, "xcds": N_XCD, "kernels": res}, fh, indent=1)
print("\n".join(lines[:16]))
