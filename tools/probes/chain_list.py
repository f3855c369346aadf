"""The generator chain of ONE serialised training step, kernel by kernel (rocprofv3 --kernel-trace CSV of tools/step_profile.py with
every side stream off): segments [acoustic model + vocoder forward] and [vocoder + acoustic model backward], each kernel's duration,
grid and the gap to its predecessor.  usage: chain_list.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:44],
              int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))) for r in rows))
# last full step: from the last-but-one text_embed_fwd to the last one
idx = [i for i, e in enumerate(ev) if e[2].startswith("text_embed_fwd")]
a, b = idx[-2], idx[-1]
step = ev[a:b]
print(f"step: {len(step)} launches, {(step[-1][1] - step[0][0]) / 1e6:.2f} ms, kernel time {sum(e - s for s, e, *_ in step) / 1e6:.2f} ms")
first_disc = next(i for i, e in enumerate(step) if e[2].startswith("period_fold") or e[2].startswith("smallcin_fwd"))
# backward of the generator below the discriminators: after the last conv_rowdot / smallcin dgrad that precedes the first layernorm_bwd / dwconv7_bwd
first_bwd = next(i for i, e in enumerate(step) if i > first_disc and (e[2].startswith("dwconv7_bwd") or e[2].startswith("layernorm_bwd")))
last_bwd = max(i for i, e in enumerate(step) if e[2].startswith("text_embed_bwd"))
def show(title, seg):
    tot = sum(e - s for s, e, *_ in seg)
    span = seg[-1][1] - seg[0][0]
    print(f"\n== {title}: {len(seg)} launches, span {span / 1e3:.0f} us, kernel time {tot / 1e3:.0f} us, gaps {(span - tot) / 1e3:.0f} us")
    agg = collections.OrderedDict()
    prev = seg[0][0]
    for s, e, n, wg in seg:
        v = agg.setdefault(n, [0, 0, 0, 0]); v[0] += 1; v[1] += e - s; v[2] += max(0, s - prev); v[3] = max(v[3], wg); prev = e
    for n, (c, t, g, wg) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {t / 1e3:8.1f} us  x{c:3d}  avg {t / c / 1e3:6.1f}  gap-before {g / 1e3:6.1f}  max WGs {wg:6d}  {n}")
show("G forward: acoustic model + vocoder", step[:first_disc])
show("G backward below the discriminators (vocoder, acoustic model)", step[first_bwd:last_bwd + 1])
