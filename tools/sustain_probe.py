#!/usr/bin/env python3
"""Does the bf16 GEMM slow down under sustained load (DVFS / power cap)?  (diagnostic)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
from optispeech_amd.disc_ops import conv1d_strided_fwd
dev = "cuda"
U, T, C = 64, 102, 1024
x = torch.randn(U, T, C, device=dev).to(torch.bfloat16)
w = K.cast_bf16(torch.randn(C, 5, C, device=dev) * 0.02)
b = torch.zeros(C, device=dev)
fl = 2.0 * U * T * 5 * C * C
for rnd in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 400
    for _ in range(n):
        conv1d_strided_fwd(x, w, b, 5, 1, 2, 0.1, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"round {rnd}: {dt*1e3:.3f} ms/call  {fl/dt/1e12:.0f} TF")
# same with randomly initialised *fresh* outputs each call vs reused allocation is identical; try zero data (DVFS check)
x.zero_()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(400):
    conv1d_strided_fwd(x, w, b, 5, 1, 2, 0.1, True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
print(f"zero activations: {dt*1e3:.3f} ms/call  {fl/dt/1e12:.0f} TF")

# per-call event timing from a cold start
import time as _t
_t.sleep(2.0)
x = torch.randn(U, T, C, device=dev).to(torch.bfloat16)
evs = []
for i in range(60):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); conv1d_strided_fwd(x, w, b, 5, 1, 2, 0.1, True); e1.record(); evs.append((e0, e1))
torch.cuda.synchronize()
print("per-call us:", " ".join(f"{a.elapsed_time(b)*1e3:.0f}" for a, b in evs))
