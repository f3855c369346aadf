#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-infer --no-am-only"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.log 2>&1; tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))" | tee -a $O/rc.txt; }
run seg1_a X=1
run seg1_b X=1
run seg1_c X=1
run seg1_d X=1
run seg1_ahead0 OSP_MAX_STEPS_AHEAD=0
run seg1_ahead1 OSP_MAX_STEPS_AHEAD=1
run seg1_ahead3 OSP_MAX_STEPS_AHEAD=3
run seg0_a OSP_TAPE_SEGMENTS=0
run seg0_b OSP_TAPE_SEGMENTS=0
