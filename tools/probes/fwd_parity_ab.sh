#!/bin/bash
# OSP_FWD_PARITY=1 (bf16 mode with the generator's training forward on the parity mode's kernels) against the default: the bf16 test
# files, the measured B = 32 errors, and same-box step times.
mkdir -p gpurun_out/fpab
for v in 1 0; do
  echo "== OSP_FWD_PARITY=$v"
  d=gpurun_out/fpab/rep$v; rm -rf $d; mkdir -p $d
  OSP_FWD_PARITY=$v OSP_TEST_REPORT=$PWD/$d timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fullsize_golden.py tests/test_gpu_tape.py tests/test_gpu_training.py tests/test_gpu_edge_cases.py -q -m gpu 2>&1 | tail -4
  for f in $d/*bf16*.txt; do echo "   $(basename $f): $(grep -E '^(am:loss|am:pitch_loss|wav_hat_l2|mr_stft|loss_g|g_am|g_voc|g_d):' $f | tr '\n' ' ')"; done
  for i in 1 2; do
  OSP_FWD_PARITY=$v timeout 300 python bench.py --precision bf16 --steps 40 --warmup 8 --no-am-only --no-infer --no-cpu-baseline --no-transformer --no-scaling-ceiling 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   bf16 ms_per_step', round(d['ms_per_step'], 3), 'host unblocked', round(d.get('host_enqueue_ms_per_step_unblocked') or 0, 2))"
  done
done
