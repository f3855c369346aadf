"""Which kernels ran on which hardware queue (rocprofv3 --kernel-trace CSV): per Queue_Id the launch count, busy time and the top symbols."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = rows[int(len(rows) * 0.4):]
q = collections.defaultdict(lambda: [0, 0, collections.Counter()])
for r in rows:
    e = q[(r.get("Queue_Id"), r.get("Stream_Id", ""))]
    e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); e[2][r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]] += 1
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span ms", span / 1e6, "launches", len(rows), "columns", [c for c in rows[0].keys() if "ueue" in c or "tream" in c])
for k, (n, busy, c) in sorted(q.items(), key=lambda kv: -kv[1][1]):
    print(k, "launches", n, "busy ms", round(busy / 1e6, 1), dict(c.most_common(4)))
