#!/bin/bash
# Everything about "two processes on one MI355X": the stand-alone HIP reproducer and the torch-only (rocFFT) one.
bash tools/probes/stft_shared_gpu.sh ${1:-2000}
echo "== torch only (rocFFT): ONE process"; timeout 300 python tools/probes/rocfft_shared_gpu.py ${2:-400} 2>&1 | grep pid
echo "== torch only (rocFFT): TWO processes side by side"
( timeout 300 python tools/probes/rocfft_shared_gpu.py ${2:-400} 2>&1 | grep pid ) > /tmp/_ra.txt & ( timeout 300 python tools/probes/rocfft_shared_gpu.py ${2:-400} 2>&1 | grep pid ) > /tmp/_rb.txt & wait
cat /tmp/_ra.txt /tmp/_rb.txt
