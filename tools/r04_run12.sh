#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-infer --no-am-only"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.log 2>&1; tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))" | tee -a $O/rc.txt; }
run burn OSP_TAPES=0
for k in 0 1 2 3; do run seg0_skewdisc$k OSP_TAPE_SEGMENTS=0 OSP_SKEW_DISC=$k; done
for k in 1 2 3; do run seg0_skewvoc$k OSP_TAPE_SEGMENTS=0 OSP_SKEW_VOCODER=$k; done
for k in 0 1 2 3; do run seg1_skewvoc$k OSP_SKEW_VOCODER=$k; done
for k in 1 2 3; do run seg1_skewdisc$k OSP_SKEW_DISC=$k; done
