#!/usr/bin/env python3
"""Does the row stride of the operands (2 KB activation rows, 10 KB weight rows: L2 channel camping?) matter for the 8-wave conv-GEMM with every CU busy?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import kernels as K
dev = "cuda"


def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


U, T, cin, n = 160, 102, 1024, 1024
M = U * T
for pa, pb in [(0, 0), (64, 0), (0, 64), (64, 64), (192, 192), (0, 0)]:
    abuf = torch.randn(U * T, cin + pa, device=dev).bfloat16()
    a = abuf[:, :cin]
    wbuf = torch.randn(n, 5, cin + pb, device=dev).bfloat16()
    w = wbuf[:, :, :cin]
    f = lambda: K.conv_gemm_bf16(a, w, n, M=M, Trows=T, Tin=T, cin=cin, taps=5, a_off=-2, out_bf16=True, lda=cin + pa,
                                 w_strides=(5 * (cin + pb), cin + pb, 1))
    t = timeit(f)
    print(f"A row stride {2 * (cin + pa)} B, weight tap stride {2 * (cin + pb)} B: {t:8.1f} us {2.0 * M * n * cin * 5 / t / 1e6:6.0f} TF", flush=True)
