#!/bin/bash
mkdir -p gpurun_out/r05d
B="python bench.py --no-cpu-baseline --no-infer --no-transformer --no-am-only --steps 60 --warmup 8"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r05d/e_$tag.json 2> gpurun_out/r05d/e_$tag.err; python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r05d/e_$tag.json').read().strip().splitlines()[-1])
    print('$tag', round(d['ms_per_step'],2), flush=True)
except Exception as e: print('$tag', 'FAILED', e)
P
}
run old OSP_EARLY_D=0 OSP_G_OPT_FIRST=0
run new A=1
run new_ts OSP_TAPE_SEGMENTS=1
run early_only OSP_G_OPT_FIRST=0
run optfirst_only OSP_EARLY_D=0
run optfirst_only_ts OSP_EARLY_D=0 OSP_TAPE_SEGMENTS=1
run old_ts OSP_EARLY_D=0 OSP_G_OPT_FIRST=0 OSP_TAPE_SEGMENTS=1
run new2 A=1
run new_ts2 OSP_TAPE_SEGMENTS=1
