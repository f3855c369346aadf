#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for p in 1 2 3; do
OSP_PIPELINE_STEPS=1 STEPS=20 rocprofv3 --kernel-trace --output-format csv -d $R/$O/p$p -o s -- python $R/tools/step_profile.py > $R/$O/p$p.log 2>&1
grep "done" $R/$O/p$p.log | tee -a $R/$O/rc.txt
python $R/tools/queue_map.py $R/$O/p$p/s_kernel_trace.csv > $R/$O/queues_p$p.txt 2>&1
python $R/tools/timeline.py $R/$O/p$p/s_kernel_trace.csv 20 > $R/$O/timeline_p$p.txt 2>&1
done
OSP_TAPE_SEGMENTS=0 OSP_PIPELINE_STEPS=1 STEPS=20 rocprofv3 --kernel-trace --output-format csv -d $R/$O/p0 -o s -- python $R/tools/step_profile.py > $R/$O/p0.log 2>&1
grep "done" $R/$O/p0.log | tee -a $R/$O/rc.txt
python $R/tools/queue_map.py $R/$O/p0/s_kernel_trace.csv > $R/$O/queues_p0.txt 2>&1
cd $R; rm -f $O/p*/*kernel_trace.csv $O/p*/*.db
for p in 1 2 0; do head -12 $O/queues_p$p.txt; head -3 $O/timeline_p$p.txt; done
