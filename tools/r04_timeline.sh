#!/bin/bash
# kernel trace of 10 steady-state steps + tools/timeline.py on it (what runs alone, idle gaps, busy time per hardware queue)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl && STEPS=10 OSP_PIPELINE_STEPS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/tools/step_profile.py > /tmp/tl.log 2>&1
tail -2 /tmp/tl.log
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline.py $f 10 > $R/gpurun_out/r04_timeline.txt 2>&1
cat $R/gpurun_out/r04_timeline.txt
