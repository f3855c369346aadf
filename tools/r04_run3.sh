#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
python tools/probes/tf_nan_probe.py > $O/tf_nan_on.log 2>&1; OSP_TAPE_SEGMENTS=0 python tools/probes/tf_nan_probe.py > $O/tf_nan_off.log 2>&1
B="python bench.py --no-cpu-baseline --no-infer --no-am-only"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.log 2>&1; tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))" | tee -a $O/rc.txt; }
run seg0 OSP_TAPE_SEGMENTS=0
run seg1 OSP_TAPE_SEGMENTS=1
run seg1_nowgradstream OSP_TAPE_SEGMENTS=1 OSP_WGRAD_STREAM=0
run seg1_novoc OSP_TAPE_SEGMENTS=1 OSP_VOC_STREAM=0
run seg1_q8 OSP_TAPE_SEGMENTS=1 GPU_MAX_HW_QUEUES=8
run seg1_q2 OSP_TAPE_SEGMENTS=1 GPU_MAX_HW_QUEUES=2
run notapes OSP_TAPES=0
B="python bench.py --no-cpu-baseline --no-infer --no-am-only --no-pipeline"
run seg1_nopipe OSP_TAPE_SEGMENTS=1
run seg0_nopipe OSP_TAPE_SEGMENTS=0
PIPE=1 OSP_TAPE_SEGMENTS=1 python tools/gpu_floor_probe.py > $O/floor_seg1.log 2>&1; PIPE=1 OSP_TAPE_SEGMENTS=0 python tools/gpu_floor_probe.py > $O/floor_seg0.log 2>&1
cd /tmp && export TMPDIR=/tmp
for s in 1 0; do
OSP_PIPELINE_STEPS=1 OSP_TAPE_SEGMENTS=$s STEPS=12 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/tl$s -o tl -- python $OLDPWD/tools/step_profile.py > $OLDPWD/$O/tl$s.log 2>&1
python $OLDPWD/tools/timeline.py $OLDPWD/$O/tl$s/tl_kernel_trace.csv 12 > $OLDPWD/$O/timeline_seg$s.txt 2>&1
done
cd $OLDPWD; rm -f $O/tl*/*kernel_trace.csv $O/tl*/*.db
cat $O/tf_nan_on.log | tail -5; cat $O/tf_nan_off.log | tail -2; tail -2 $O/floor_seg1.log $O/floor_seg0.log; cat $O/timeline_seg1.txt | head -12; cat $O/timeline_seg0.txt | head -12
