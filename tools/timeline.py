"""Timeline of the eager multi-stream step from a rocprofv3 --kernel-trace CSV: per step the span, the union of busy time, the
idle gaps and the time during which only small (< 64 workgroup) kernels were running.  usage: timeline.py <kernel_trace.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2])
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * (int(r.get("Grid_Size_Y", 1) or 1) // max(1, int(r.get("Workgroup_Size_Y", 1) or 1))) * (int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r.get("Workgroup_Size_Z", 1) or 1)))
    ev.append((s, e, r["Kernel_Name"].split("(")[0][:50], wg, r.get("Queue_Id", "0")))
ev.sort()
# steady state: drop the first 40 % (set-up, warm-up)
t0 = ev[int(len(ev) * 0.4)][0]
ev = [x for x in ev if x[0] >= t0]
span = ev[-1][1] - ev[0][0]
# union of intervals
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
for s, e, *_ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{len(ev)} launches over {span/1e6:.2f} ms; union busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), idle {100*(span-busy)/span:.1f} %; sum of kernel durations {sum(e-s for s,e,*_ in ev)/1e6:.2f} ms")
gaps.sort(reverse=True)
print("largest idle gaps (us):", [round(g / 1e3, 1) for g, _ in gaps[:12]], " #gaps > 5us:", sum(1 for g, _ in gaps if g > 5000), " total in gaps>5us (ms):", round(sum(g for g, _ in gaps if g > 5000) / 1e6, 2))
# concurrency-weighted: time with exactly one kernel running, by kernel
pts = []
for i, (s, e, *_r) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
active, last, alone = set(), pts[0][0], {}
for t, d, i in pts:
    if len(active) == 1:
        k = ev[next(iter(active))][2]
        alone[k] = alone.get(k, 0) + (t - last)
    last = t
    if d == 1: active.add(i)
    else: active.discard(i)
tot_alone = sum(alone.values())
print(f"time with exactly ONE kernel in flight: {tot_alone/1e6:.2f} ms ({100*tot_alone/span:.1f} % of the span); by kernel:")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
    print(f"   {v/1e6:7.2f} ms  {k}")
qs = {}
for s, e, k, wg, q in ev:
    qs[q] = qs.get(q, 0) + (e - s)
print("busy time per queue (ms):", {q: round(v / 1e6, 1) for q, v in sorted(qs.items(), key=lambda kv: -kv[1])})
# context of the largest idle gaps: the kernels that end just before and start just after
ends = sorted((e, k) for s, e, k, *_ in ev)
starts = sorted((s, k) for s, e, k, *_ in ev)
import bisect
print("largest gaps: [last kernels to finish] -> gap -> [first kernels to start]")
for g, at in gaps[:6]:
    i = bisect.bisect_right(ends, (at, "~")) 
    j = bisect.bisect_left(starts, (at + g, ""))
    print(f"  {g/1e3:7.1f} us  after {[k for _, k in ends[max(0, i-3):i]]}  before {[k for _, k in starts[j:j+3]]}")
