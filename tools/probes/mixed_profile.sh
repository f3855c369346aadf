cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mp && PRECISION=mixed OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -o mp -- python $R/tools/step_profile.py > /tmp/mp.log 2>&1
tail -1 /tmp/mp.log
python - <<PY
import csv,glob
f=glob.glob("/tmp/mp/**/mp_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms/step:", tot/13e6)
for r in rows[:26]: print("%7.2f ms/step %4d x %7.1f us  %s" % (float(r["TotalDurationNs"])/13e6, int(r["Calls"])//13, float(r["AverageNs"])/1e3, r["Name"][:80]))
PY
