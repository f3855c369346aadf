#!/usr/bin/env python3
"""Fixed cost of the 8-wave weight-gradient kernel: the MPD conv4 shape (1024 -> 1024, 5 taps) at a growing number of frames
(the launch always covers 80 tiles x its frame splits; small M = prologue + atomic epilogue only).  OSP_WGRAD_W8_TARGET=80 runs it
without frame splits (plain read-modify-write epilogue instead of atomics)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
C, T = 1024, 102
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("OSP_WGRAD_W8_TARGET =", os.environ.get("OSP_WGRAD_W8_TARGET"))
for (cout, cin) in ((1024, 1024), (1024, 512)):
    for U in (8, 16, 32, 64, 128):
        M = U * T
        dy = torch.randn(U, 1, T, cout, device=dev).to(torch.bfloat16)
        x = torch.randn(U, 1, T, cin, device=dev).to(torch.bfloat16)
        dw = torch.zeros(cout, 1, 5, cin, device=dev); db = torch.zeros(cout, device=dev)
        f = lambda: K.conv2d_wgrad_bf16(dy.view(M, cout), x.view(M, cin), dw, db, M=M, Trows=T, Wrows=T, Hin=1, Win=T, n=cout, cin=cin, taps=5, KW=5, pad_h=0, pad_w=2, step_h=1, step_w=1)
        us = t(f)
        print(f"{cin:4d}->{cout:4d} M={M:6d}: {us:7.1f} us  {2.0*M*5*cin*cout/us/1e6:6.0f} TF")
