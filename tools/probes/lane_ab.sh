#!/bin/bash
# A/B of stream -> hardware-queue tables with the generator chain on an explicit "main" stream (same box, bench.py timed region only)
mkdir -p gpurun_out/r05d
B="python bench.py --no-cpu-baseline --no-infer --no-transformer --no-am-only --steps 60 --warmup 8"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r05d/l_$tag.json 2> gpurun_out/r05d/l_$tag.err; python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r05d/l_$tag.json').read().strip().splitlines()[-1])
    print('$tag', round(d['ms_per_step'],2), flush=True)
except Exception as e: print('$tag', 'FAILED', e)
P
}
TS="OSP_TAPE_SEGMENTS=1"
run base_eager A=1
run base_ts $TS
# generator chain alone on lane 0, discriminator work on lanes 1-3
run m0_d123 $TS OSP_LANES="main:0,voc:0,wg_voc:0,wg_main:0,ctc:1,spec:1,dphase:1,p0:1,p1:2,p2:3,p3:1,p4:2,r0:3,r1:2,r2:3"
run m0_d123_b $TS OSP_LANES="main:0,voc:0,wg_voc:1,wg_main:2,ctc:3,spec:1,dphase:1,p0:1,p1:2,p2:3,p3:1,p4:2,r0:3,r1:2,r2:3"
run m0_d123_c $TS OSP_LANES="main:0,voc:0,wg_voc:0,wg_main:0,ctc:0,spec:0,dphase:1,p0:1,p1:2,p2:3,p3:1,p4:2,r0:3,r1:2,r2:3"
# the round-4 table with main made explicit on each lane
run r4_m0 $TS OSP_LANES="main:0"
run r4_m1 $TS OSP_LANES="main:1"
run r4_m2 $TS OSP_LANES="main:2"
run r4_m3 $TS OSP_LANES="main:3"
# generator on 2 lanes (main 0, voc 1), discriminators on 2-3 + shared
run m0v1 $TS OSP_LANES="main:0,voc:1,wg_voc:1,wg_main:0,ctc:1,spec:1,dphase:2,p0:2,p1:3,p2:2,p3:3,p4:2,r0:3,r1:2,r2:3"
run m0_d123_eager OSP_LANES="main:0,voc:0,wg_voc:0,wg_main:0,ctc:1,spec:1,dphase:1,p0:1,p1:2,p2:3,p3:1,p4:2,r0:3,r1:2,r2:3"
run base_ts2 $TS
