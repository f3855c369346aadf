#!/bin/bash
# usage (on the GPU box): bash tools/probes/stft_shared_gpu.sh [iters]  -- prints one RESULT line per process
BIN=tools/probes/stft_shared_gpu
IT=${1:-2000}
for v in stft wave ldsmix nolds; do
  echo "== $v: ONE process"; timeout 300 $BIN $v $IT | grep -v "^  " | tail -4
  echo "== $v: TWO processes side by side"
  ( timeout 300 $BIN $v $IT | sed "s/^/A: /" ) > /tmp/_sa.txt & ( timeout 300 $BIN $v $IT | sed "s/^/B: /" ) > /tmp/_sb.txt & wait
  grep "RESULT\|deviate" /tmp/_sa.txt /tmp/_sb.txt | cut -d: -f2-
  echo "== $v next to a HOG process (long kernels on every CU, 64 KB LDS per workgroup: the victim only runs by time-slicing)"
  ( timeout 300 $BIN hog 20000 ) > /tmp/_sh.txt & HP=$!
  sleep 1
  timeout 300 $BIN $v $IT | grep -v "^  " | tail -4
  kill $HP 2>/dev/null; wait $HP 2>/dev/null; cat /tmp/_sh.txt
done
echo "#### both processes HEAVY: each interleaves a long all-CU LDS kernel with the victim kernel (OSP_PROBE_MIX)"
for v in stft wave ldsmix nolds; do
  echo "== $v, heavy, ONE process"; OSP_PROBE_MIX=40000 timeout 600 $BIN $v 4000 | grep -v "^  " | tail -4
  echo "== $v, heavy, TWO processes side by side"
  ( OSP_PROBE_MIX=40000 timeout 600 $BIN $v 4000 | sed "s/^/A: /" ) > /tmp/_sa.txt & ( OSP_PROBE_MIX=40000 timeout 600 $BIN $v 4000 | sed "s/^/B: /" ) > /tmp/_sb.txt & wait
  grep "RESULT\|deviate" /tmp/_sa.txt /tmp/_sb.txt | cut -d: -f2-
done
echo "#### the same with 144 KB of dynamic LDS in the heavy kernel (the 8-wave conv-GEMM's allocation)"
for v in stft ldsmix; do
  echo "== $v, heavy 144 KB, TWO processes side by side"
  ( OSP_PROBE_HOG_LDS=147456 OSP_PROBE_MIX=40000 timeout 600 $BIN $v 4000 | sed "s/^/A: /" ) > /tmp/_sa.txt & ( OSP_PROBE_HOG_LDS=147456 OSP_PROBE_MIX=40000 timeout 600 $BIN $v 4000 | sed "s/^/B: /" ) > /tmp/_sb.txt & wait
  grep "RESULT\|deviate" /tmp/_sa.txt /tmp/_sb.txt | cut -d: -f2-
done
