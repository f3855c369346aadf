#!/usr/bin/env python3
"""glds conv-GEMM kernel with the load / MFMA halves disabled (OSP_GEMM_DBG=1: no loads, 2: no MFMA) -- diagnostic."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for (M, N, Kd) in [(13056, 1024, 5120), (6528, 1024, 5120), (16384, 4096, 4096)]:
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    w = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    fl = 2.0 * M * N * Kd
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tm = t(lambda: K.conv_gemm_bf16(a, w, N, M=M, Trows=M, Tin=M, cin=Kd, out=out, out_bf16=True))
    print(f"DBG={os.environ.get('OSP_GEMM_DBG','0')} M={M} N={N} K={Kd}: {tm*1e3:.1f} us ({fl/tm/1e9:.0f} TF)")
