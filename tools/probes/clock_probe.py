#!/usr/bin/env python3
"""Engine clock / socket power while one GEMM shape runs back to back for a few seconds (rocm-smi sampled from a second thread):
is the 8-wave conv-GEMM's per-tile slowdown with every CU busy a clock / power-cap effect?"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import kernels as K
dev = "cuda"
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.perf_counter(), out))
        except Exception as e:                               # noqa: BLE001
            samples.append((time.perf_counter(), f"ERR {e}"))
        time.sleep(0.2)


def run(tag, f, secs=4.0):
    global samples
    f(); torch.cuda.synchronize()
    samples = []
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(200):
            f()
        torch.cuda.synchronize(); n += 200
    dt = (time.perf_counter() - t0) / n * 1e6
    mid = [s for t, s in samples if t0 + 1.0 < t < t0 + secs]
    print(f"== {tag}: {dt:.1f} us / launch; {len(mid)} samples", flush=True)
    if mid:
        print(mid[len(mid) // 2], flush=True)


th = threading.Thread(target=sampler, daemon=True); th.start()
w = torch.randn(1024, 5, 1024, device=dev).bfloat16()
for U in (64, 160):
    M = U * 102
    a = torch.randn(M, 1024, device=dev).bfloat16()
    run(f"conv-GEMM 8-wave, {M // 256 * 4 if M % 256 == 0 else (M // 256 + 1) * 4} tiles", lambda: K.conv_gemm_bf16(a, w, 1024, M=M, Trows=102, Tin=102, cin=1024, taps=5, a_off=-2, out_bf16=True))
w2 = torch.randn(1024, 5120, device=dev).bfloat16()
for M in (6528, 16320):
    a2 = torch.randn(M, 5120, device=dev).bfloat16()
    run(f"hipBLASLt M={M}", lambda: torch.matmul(a2, w2.t()))
stop = True
