#!/bin/bash
# headline (bf16) and parity-mode (mixed) step times with the secondary figures off; usage: quick_ab.sh "ENV=.. ENV=.." ...  (one line per configuration)
mkdir -p gpurun_out/quick
run() {
  env $1 timeout 300 python bench.py --precision $2 --steps 30 --warmup 8 --no-am-only --no-infer --no-cpu-baseline --no-transformer --no-scaling-ceiling 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   $2 ms_per_step', round(d['ms_per_step'], 3))"
}
for cfg in "$@"; do
  echo "== $cfg"
  run "$cfg" bf16
  run "$cfg" mixed
done
