#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
smi() { rocm-smi --showclocks --showtemp --showpower 2>/dev/null | grep -i "sclk\|mclk\|Temperature (Sensor junction)\|Average Graphics Package Power\|Current Socket" | head -6 | tr '\n' ';'; echo; }
B="python bench.py --no-cpu-baseline --no-infer --no-am-only"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.log 2>&1; tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))" | tee -a $O/rc.txt; smi | tee -a $O/rc.txt; }
smi | tee $O/rc.txt
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
OSP_PIPELINE_STEPS=1 STEPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p1 -o s -- python $R/tools/step_profile.py > $R/$O/p1.log 2>&1
OSP_PIPELINE_STEPS=1 STEPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p2 -o s -- python $R/tools/step_profile.py > $R/$O/p2.log 2>&1
cd $R
rm -f $O/p*/*kernel_trace.csv $O/p*/*.db
run seg1_a X=1
run seg1_b X=1
sleep 30
run seg1_after_sleep X=1
run seg0 OSP_TAPE_SEGMENTS=0
run seg1_c X=1
ls /dev/shm | head; 
