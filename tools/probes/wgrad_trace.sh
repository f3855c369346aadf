#!/bin/bash
# per-kernel durations of the weight-gradient probe: tools/probes/wgrad_trace.sh <tag> [env...]
TAG=$1; shift
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" REP=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o tr -- python $R/tools/probes/wgrad_shapes.py > $O/trace.log 2>&1
cd $R
cp $O/tr/*/tr_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/tr/tr_kernel_stats.csv $O/kernel_stats.csv
python - "$O" <<'P'
import csv, glob, sys, os, collections
o = sys.argv[1]
f = glob.glob(os.path.join(o, "tr", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open(os.path.join(o, "launches.txt"), "w")
for r in rows:
    n = r["Kernel_Name"]
    if "wgrad" in n:
        out.write(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  grid {r.get("Grid_Size_X", "?"):>8s}  {n[:70]}\n')
P
rm -rf $O/tr
