mkdir -p gpurun_out/hostexp
F="--steps 40 --warmup 10 --no-cpu-baseline --no-am-only --no-infer"
run() { label=$1; shift; for r in 1 2; do env "$@" python bench.py $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3), 'enq', round(d['host_enqueue_ms_per_step'],3))"; done; }
run default X=1
run gc-off OSP_GC_OFF=1
run autograd-st OSP_AUTOGRAD_ST=1
run both OSP_GC_OFF=1 OSP_AUTOGRAD_ST=1
run default X=1
