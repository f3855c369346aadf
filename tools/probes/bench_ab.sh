mkdir -p gpurun_out/r05b
B="python bench.py --no-cpu-baseline --no-infer --no-transformer --no-am-only"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r05b/b_$tag.json 2> gpurun_out/r05b/b_$tag.err; python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r05b/b_$tag.json').read().strip().splitlines()[-1])
    print('$tag', round(d['ms_per_step'],2), 'host', round(d.get('host_enqueue_ms_per_step') or 0,2), 'unblocked', d.get('host_enqueue_ms_per_step_unblocked'))
except Exception as e: print('$tag', 'FAILED', e)
P
}
run default A=1
run tapeseg OSP_TAPE_SEGMENTS=1
run tapeseg_am OSP_TAPE_SEGMENTS=1 OSP_TAPE_VOC=0
run tapeseg_voc OSP_TAPE_SEGMENTS=1 OSP_TAPE_AM=0
run tapeseg_ahead3 OSP_TAPE_SEGMENTS=1 OSP_MAX_STEPS_AHEAD=3
run tapeseg_ahead1 OSP_TAPE_SEGMENTS=1 OSP_MAX_STEPS_AHEAD=1
run default2 A=1
run tapeseg2 OSP_TAPE_SEGMENTS=1
