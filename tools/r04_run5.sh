#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
OSP_TAPE_SEGMENTS=0 python tools/probes/tf_nan_probe.py 2>&1 | grep BISECT | tee -a $O/rc.txt
python tools/probes/tf_nan_probe.py 2>&1 | grep BISECT | tee -a $O/rc.txt
B="python bench.py --no-cpu-baseline --no-infer --no-am-only"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.log 2>&1; tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))" | tee -a $O/rc.txt; }
run seg1 OSP_TAPE_SEGMENTS=1
run seg1_pace2 OSP_TAPE_PACE_NS=2000
run seg1_pace5 OSP_TAPE_PACE_NS=5000
run seg1_pace10 OSP_TAPE_PACE_NS=10000
run seg1_pace15 OSP_TAPE_PACE_NS=15000
run seg1_pace25 OSP_TAPE_PACE_NS=25000
run seg0 OSP_TAPE_SEGMENTS=0
run seg0_pace10 OSP_TAPE_SEGMENTS=0 OSP_TAPE_PACE_NS=10000
