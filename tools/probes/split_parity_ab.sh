#!/bin/bash
# Which split-bf16 call sites move the B = 32 gradient-norm checks (tests/test_gpu_fullsize_golden.py, precision "mixed"), and what each
# configuration's step costs: usage split_parity_ab.sh "ENV=.." "ENV=.." ...   ("-" = no extra environment)
mkdir -p gpurun_out/spab
for cfg in "$@"; do
  e=$cfg; [ "$cfg" = "-" ] && e="OSP_NOP=1"
  d=gpurun_out/spab/$(echo "$cfg" | tr -c 'A-Za-z0-9=_\n' '_')
  rm -rf $d; mkdir -p $d
  env $e OSP_TEST_REPORT=$PWD/$d timeout 300 python -m pytest tests/test_gpu_fullsize_golden.py -q -m gpu -k mixed 2>&1 | tail -1
  echo "== $cfg"
  for f in $d/*.txt; do echo "   $(basename $f): $(grep -E '^(wav_hat_l2|g_am|g_voc|g_d):' $f | tr '\n' ' ')"; done
  env $e timeout 300 python bench.py --precision mixed --steps 30 --warmup 8 --no-am-only --no-infer --no-cpu-baseline --no-transformer --no-scaling-ceiling 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   mixed ms_per_step', round(d['ms_per_step'], 3))"
done
