#!/bin/bash
# kernel trace of 10 steady-state steps + tools/timeline.py on it: tools/r05_timeline.sh <tag> [ENV=VAL ...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05t
rm -rf /tmp/tl_$TAG && env "$@" STEPS=10 OSP_PIPELINE_STEPS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -- python $R/tools/step_profile.py > /tmp/tl_$TAG.log 2>&1
tail -2 /tmp/tl_$TAG.log
f=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline.py $f 10 > $R/gpurun_out/r05t/timeline_$TAG.txt 2>&1
head -24 $R/gpurun_out/r05t/timeline_$TAG.txt
