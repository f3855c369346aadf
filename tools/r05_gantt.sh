#!/bin/bash
# tools/r05_gantt.sh <tag> [ENV=VAL ...]: kernel trace of 12 steps -> gantt of the last two
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05t
rm -rf /tmp/gt_$TAG && env "$@" STEPS=12 OSP_PIPELINE_STEPS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt_$TAG -- python $R/tools/step_profile.py > /tmp/gt_$TAG.log 2>&1
tail -1 /tmp/gt_$TAG.log
f=$(find /tmp/gt_$TAG -name "*kernel_trace.csv" | head -1)
python $R/tools/gantt.py $f 250 34 > $R/gpurun_out/r05t/gantt_$TAG.txt 2>&1
python $R/tools/timeline.py $f 12 > $R/gpurun_out/r05t/timeline_$TAG.txt 2>&1
