#!/bin/bash
# A/B of bench.py step time under env variants: ab.sh "<env1>" "<env2>" ... ; two rounds interleaved
mkdir -p gpurun_out/ab
for r in 1 2; do
for i in "$@"; do
  tag=$(echo "$i" | tr ' =' '__')
  env $i python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-am-only --no-infer --no-transformer --no-scaling-ceiling 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$i', 'round $r', 'ms_per_step', round(d['ms_per_step'],3), 'roofline', d['roofline']['symbol'], round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],1))
"
done
done
