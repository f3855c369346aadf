#!/bin/bash
mkdir -p gpurun_out/r05d
B="python bench.py --no-cpu-baseline --no-infer --no-transformer --no-am-only --steps 60 --warmup 8"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r05d/f_$tag.json 2> gpurun_out/r05d/f_$tag.err; python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r05d/f_$tag.json').read().strip().splitlines()[-1])
    print('$tag', round(d['ms_per_step'],2), flush=True)
except Exception as e: print('$tag', 'FAILED', e)
P
}
TS=OSP_TAPE_SEGMENTS=1
run old_ts OSP_EARLY_D=0 OSP_G_OPT_FIRST=0 $TS
for l in 0 2 3; do
run new_ts_d$l $TS OSP_LANES="dphase:$l"
run early_ts_d$l $TS OSP_G_OPT_FIRST=0 OSP_LANES="dphase:$l"
run old_ts_d$l $TS OSP_EARLY_D=0 OSP_G_OPT_FIRST=0 OSP_LANES="dphase:$l"
done
run new_ts_dfree $TS OSP_LANES="dphase:-"
run early_ts_dfree $TS OSP_G_OPT_FIRST=0 OSP_LANES="dphase:-"
