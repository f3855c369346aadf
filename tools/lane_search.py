#!/usr/bin/env python3
"""Search the stream -> hardware-queue assignment (optispeech_amd/lanes.py): runs bench.py once per candidate OSP_LANES setting and
prints ms / step, best first.   usage: lane_search.py <out dir> <n random> [seed] [extra env k=v ...]"""
import json, os, random, subprocess, sys
out, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0
extra = dict(kv.split("=", 1) for kv in sys.argv[4:])
names = ["voc", "ctc", "wg_main", "wg_voc", "p0", "p1", "p2", "p3", "p4", "r0", "r1", "r2", "spec", "dphase"]
rnd = random.Random(seed)
NL = int(extra.get("OSP_N_LANES", "4"))
cands = [("unmanaged", "")]
fixed = os.environ.get("CANDS")
if fixed:
    for i, c in enumerate(fixed.split(";")):
        cands.append((f"given{i}", c))
base = os.environ.get("BASE")
if base:
    b = dict(kv.split(":") for kv in base.split(","))
    cands.append(("base", base)); cands.append(("base_again", base))
    for i in range(n):
        a = dict(b)
        for nm in rnd.sample(names, rnd.choice([1, 1, 2])):
            a[nm] = str(rnd.randrange(NL))
        cands.append((f"mut{i}", ",".join(f"{k}:{v}" for k, v in a.items())))
    n = 0
for i in range(n):
    # the eight stacks: a random balanced deal (two per lane); everything else uniformly random
    deal = [k % NL for k in range(8)]
    rnd.shuffle(deal)
    a = {nm: rnd.randrange(NL) for nm in names}
    for nm, l in zip(["p0", "p1", "p2", "p3", "p4", "r0", "r1", "r2"], deal):
        a[nm] = l
    cands.append((f"rand{i}", ",".join(f"{k}:{v}" for k, v in a.items())))
os.makedirs(out, exist_ok=True)
res = []
for tag, lanes in cands:
    env = dict(os.environ, OSP_LANES=lanes, **extra)
    if tag == "unmanaged":
        env["OSP_LANES_OFF"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-infer", "--no-am-only"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        res.append((d["ms_per_step"], d["host_enqueue_ms_per_step"], tag, lanes))
        print(f"{tag:10s} {d['ms_per_step']:.2f} ms/step host {d['host_enqueue_ms_per_step']:.2f}  {lanes}", flush=True)
    except Exception as e:
        print(tag, "FAILED", r.stderr[-300:], flush=True)
res.sort()
with open(os.path.join(out, "lane_search.txt"), "a") as fh:
    fh.write(f"# extra env {extra}\n")
    for ms, host, tag, lanes in res:
        fh.write(f"{ms:.2f} {host:.2f} {tag} {lanes}\n")
print("BEST:", res[:5])
