#!/bin/bash
# kernel statistics of synthesise() (64 sentences, 10 repetitions): gpurun_out/r04_synth/synth_kernel_stats.csv
R=$PWD; O=$R/gpurun_out/r04_synth; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o synth -- python $R/tools/synth_profile.py > $O/synth.log 2>&1
rm -f $O/*_kernel_trace.csv $O/*.db
tail -3 $O/synth.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/synth_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per call (13 calls):", tot/13e6)
for r in rows[:22]:
    print("%8.1f us/call %5d calls/call avg %7.1f us  %s" % (float(r["TotalDurationNs"])/13e3, int(r["Calls"])//13, float(r["AverageNs"])/1e3, r["Name"][:90]))
PY
