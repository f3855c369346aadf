#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
for b in "" scale_dev mul_rows relu_mask axpby sum_scaled dot masks transpose; do BISECT=$b OSP_TAPE_SEGMENTS=0 python tools/probes/tf_nan_probe.py 2>&1 | grep BISECT | tee -a $O/bisect.txt; done
B="python bench.py --no-cpu-baseline --no-infer --no-am-only"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.log 2>&1; tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))" | tee -a $O/rc.txt; }
run ahead0 OSP_MAX_STEPS_AHEAD=0
run ahead1 OSP_MAX_STEPS_AHEAD=1
run ahead2 OSP_MAX_STEPS_AHEAD=2
run ahead3 OSP_MAX_STEPS_AHEAD=3
run ahead1_seg0 OSP_MAX_STEPS_AHEAD=1 OSP_TAPE_SEGMENTS=0
run ahead2_seg0 OSP_MAX_STEPS_AHEAD=2 OSP_TAPE_SEGMENTS=0
