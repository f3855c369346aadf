#!/bin/bash
# precision "mixed" step time: this round's f32 ring weight gradients / tape-safe f32 packs against round 4's kernels, one box
mkdir -p gpurun_out/mixed
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "== OSP_WGRAD_RING_F32=$1 OSP_F32_PACKS_UNDER_TAPE=$2"
  OSP_WGRAD_RING_F32=$1 OSP_F32_PACKS_UNDER_TAPE=$2 timeout 300 python bench.py --precision mixed --steps 20 --warmup 5 --no-am-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'], 3))"
done
