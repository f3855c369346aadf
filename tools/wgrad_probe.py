#!/usr/bin/env python3
"""Single-kernel driver for PMC runs: the MPD conv4 weight-gradient GEMM (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
U, T, C = 128, 102, 1024
dy = torch.randn(U, 1, T, C, device=dev).to(torch.bfloat16)
x = torch.randn(U, 1, T, C, device=dev).to(torch.bfloat16)
dw = torch.zeros(C, 1, 5, C, device=dev); db = torch.zeros(C, device=dev)
def f():
    K.conv2d_wgrad_bf16(dy.view(U * T, C), x.view(U * T, C), dw, db, M=U * T, Trows=T, Wrows=T, Hin=1, Win=T, n=C, cin=C, taps=5, KW=5, pad_h=0, pad_w=2, step_h=1, step_w=1)
for _ in range(3): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"wgrad {dt*1e3:.3f} ms  {2.0*U*T*5*C*C/dt/1e12:.0f} TF")
