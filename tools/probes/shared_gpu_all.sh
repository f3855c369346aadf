#!/bin/bash
# Everything about "two processes on one MI355X": the stand-alone HIP reproducer and the torch-only (rocFFT) one.
bash tools/probes/stft_shared_gpu.sh ${1:-2000}
echo "== torch only (rocFFT): ONE process"; timeout 300 python tools/probes/rocfft_shared_gpu.py ${2:-400} 2>&1 | grep pid
echo "== torch only (rocFFT): TWO processes side by side"
( timeout 300 python tools/probes/rocfft_shared_gpu.py ${2:-400} 2>&1 | grep pid ) > /tmp/_ra.txt & ( timeout 300 python tools/probes/rocfft_shared_gpu.py ${2:-400} 2>&1 | grep pid ) > /tmp/_rb.txt & wait
cat /tmp/_ra.txt /tmp/_rb.txt
echo "== torch only (rocFFT) next to a HOG process"
( timeout 300 tools/probes/stft_shared_gpu hog 60000 ) > /tmp/_sh.txt & HP=$!
sleep 1
timeout 300 python tools/probes/rocfft_shared_gpu.py ${2:-400} 2>&1 | grep pid
kill $HP 2>/dev/null; wait $HP 2>/dev/null; cat /tmp/_sh.txt
echo "== torch only (rocFFT) next to the repository's own training step (bench.py loop) in a second process"
( timeout 200 python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-infer --no-am-only > /tmp/_bench.txt 2>&1 ) & BP=$!
sleep 25
timeout 300 python tools/probes/rocfft_shared_gpu.py ${2:-400} 2>&1 | grep pid
timeout 300 tools/probes/stft_shared_gpu stft ${1:-2000} | grep -v "^  " | tail -4
wait $BP 2>/dev/null; tail -c 300 /tmp/_bench.txt
