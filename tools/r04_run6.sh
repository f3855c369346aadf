#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" 2>&1 | tail -1 | tee $O/rc.txt
B="python bench.py --no-cpu-baseline --no-infer --no-am-only"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.log 2>&1; tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))" | tee -a $O/rc.txt; }
run seg1 X=1
run seg1_chain_hi OSP_PRIO_CHAIN=-1
run seg1_disc_lo OSP_PRIO_DISC=1
run seg1_disc_lo_chain_hi OSP_PRIO_DISC=1 OSP_PRIO_CHAIN=-1
run seg1_d_after_g OSP_D_AFTER_G=1
run seg0 OSP_TAPE_SEGMENTS=0
run seg0_d_after_g OSP_TAPE_SEGMENTS=0 OSP_D_AFTER_G=1
