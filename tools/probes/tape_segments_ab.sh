#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-infer --no-am-only"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.log 2>&1; tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))" | tee -a $O/rc.txt; }
run burn OSP_TAPES=0
run am_only_1 OSP_TAPE_VOC=0
run am_only_2 OSP_TAPE_VOC=0
run voc_only_1 OSP_TAPE_AM=0
run voc_only_2 OSP_TAPE_AM=0
run both_1 X=1
run none_1 OSP_TAPE_AM=0 OSP_TAPE_VOC=0
run both_wgoff OSP_WGRAD_STREAM=0 OSP_VOC_STREAM=0
run both_2 X=1
run seg0 OSP_TAPE_SEGMENTS=0
