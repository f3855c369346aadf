"""Summarise a rocprofv3 kernel_stats.csv of tools/step_profile.py per training step: python tools/stats_per_step.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / n
calls = sum(int(r["Calls"]) for r in rows) / n
print(f"kernel time {tot:.2f} ms/step, {calls:.0f} launches/step")
acc = 0
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    ms = float(r["TotalDurationNs"]) / 1e6 / n
    acc += ms
    print(f"{ms:7.3f} ms {int(r['Calls'])/n:7.1f} x {float(r['AverageNs'])/1e3:8.1f} us  cum {acc:6.2f}  {r['Name'][:120]}")
