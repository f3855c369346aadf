#!/usr/bin/env python3
"""bf16 conv-GEMM kernel vs torch.matmul (hipBLASLt) on plain GEMM shapes (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (M, N, Kd) in [(6688, 1024, 5120), (8192, 1024, 1024), (25600, 1024, 256), (25600, 256, 1024), (16384, 4096, 4096)]:
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    w = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    fl = 2.0 * M * N * Kd
    tm = t(lambda: K.conv_gemm_bf16(a, w, N, M=M, Trows=M, Tin=M, cin=Kd, out_bf16=True))
    tt = t(lambda: torch.matmul(a, w.t()))
    af, wf = a.float(), w.float()
    t32 = t(lambda: K.conv_gemm(af, wf, N))
    print(f"M={M} N={N} K={Kd}: mine bf16 {tm:.3f} ms ({fl/tm/1e9:.0f} TF) | torch bf16 {tt:.3f} ms ({fl/tt/1e9:.0f} TF) | mine f32 {t32:.3f} ms ({fl/t32/1e9:.0f} TF)")
