#!/usr/bin/env python3
"""Second, finer search of the stream -> hardware-queue table (optispeech_amd/lanes.py) on ONE box: tools/step_profile.py (80 pipelined
steps) per candidate -- mutations of the current default and random balanced deals --, then the best few re-measured three times next
to the default.   usage: lane_search2.py <out file> <n mutations> <n random> [seed]"""
import os, random, re, subprocess, sys
out, nmut, nrand = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rnd = random.Random(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
names = ["voc", "ctc", "wg_main", "wg_voc", "p0", "p1", "p2", "p3", "p4", "r0", "r1", "r2", "spec", "dphase"]
DEFAULT = dict(voc=1, ctc=1, wg_main=3, wg_voc=0, p0=3, p1=1, p2=2, p3=0, p4=3, r0=1, r1=0, r2=2, spec=0, dphase=1)   # round-4 table before this search
BASE = dict(DEFAULT)
if os.environ.get("BASE"):                                  # mutate around another table: BASE="voc:0,ctc:1,..."
    BASE.update({k: int(v) for k, v in (kv.split(":") for kv in os.environ["BASE"].split(","))})
fmt = lambda a: ",".join(f"{k}:{a[k]}" for k in names)


def measure(lanes, steps=80):
    env = dict(os.environ, OSP_LANES=lanes, STEPS=str(steps), OSP_PIPELINE_STEPS="1")
    r = subprocess.run([sys.executable, "tools/step_profile.py"], env=env, capture_output=True, text=True)
    m = re.search(r"([0-9.]+) ms per step", r.stdout)
    return float(m.group(1)) if m else float("nan")


cands = [("base", fmt(BASE))]
for i in range(nmut):
    a = dict(BASE)
    for nm in rnd.sample(names, rnd.choice([1, 1, 2, 3])):
        a[nm] = rnd.randrange(4)
    cands.append((f"mut{i}", fmt(a)))
for i in range(nrand):
    deal = [k % 4 for k in range(8)]
    rnd.shuffle(deal)
    a = {nm: rnd.randrange(4) for nm in names}
    for nm, l in zip(["p0", "p1", "p2", "p3", "p4", "r0", "r1", "r2"], deal):
        a[nm] = l
    cands.append((f"rand{i}", fmt(a)))
res = []
with open(out, "w") as fh:
    fh.write("# python tools/lane_search2.py: 80 pipelined steps per candidate (tools/step_profile.py), one box\n")
    for tag, lanes in cands:
        ms = measure(lanes)
        res.append((ms, tag, lanes))
        line = f"{ms:.2f} {tag} {lanes}"
        print(line, flush=True); fh.write(line + "\n"); fh.flush()
    res.sort()
    fh.write("# re-measured (3 x 120 steps each, interleaved with the default)\n")
    for ms, tag, lanes in [r for r in res[:6]] + [(0, "base", fmt(BASE)), (0, "default", fmt(DEFAULT))]:
        xs = [measure(lanes, 120) for _ in range(3)]
        line = f"{min(xs):.2f} {sorted(xs)[1]:.2f} {max(xs):.2f}  {tag} {lanes}"
        print("RE", line, flush=True); fh.write(line + "\n"); fh.flush()
