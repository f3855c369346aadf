# A/B of schedule knobs on the default bench configuration (two runs each; box-to-box noise is ~0.1 ms): bash tools/hostexp.sh
mkdir -p gpurun_out/hostexp
F="--steps 30 --warmup 8 --no-cpu-baseline --no-am-only --no-infer"
run() { label=$1; shift; for r in 1 2; do env "$@" python bench.py $F $EXTRA 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3), 'enq', round(d['host_enqueue_ms_per_step'],3))"; done; }
run default X=1
run flush2 OSP_WGRAD_FLUSH=2
run flush4 OSP_WGRAD_FLUSH=4
run wgrad-inline OSP_WGRAD_STREAM=0
run disc-streams-4 OSP_DISC_MAX_STREAMS=4
run disc-streams-6 OSP_DISC_MAX_STREAMS=6
run no-voc-stream OSP_VOC_STREAM=0
run gc-off OSP_GC_OFF=1
EXTRA=--no-pipeline run no-pipeline X=1
run default X=1
