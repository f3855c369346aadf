#!/usr/bin/env python3
"""8-wave 256x256 conv-GEMM vs the 4-wave 128x128 one on the DiscriminatorP layer shapes (forward + fused-phase dgrad) and on
plain GEMMs.  Run as:  OSP_GEMM_W8=0 python tools/gemm_w8_probe.py ; OSP_GEMM_W8=1 python tools/gemm_w8_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import disc_ops as D, kernels as K, precision
precision.set_precision("bf16")
dev = "cuda"


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tot = 0.0
print("W8 =", os.environ.get("OSP_GEMM_W8"))
for half in (1, 2):
    for p in (2, 3, 5, 7, 11):
        U = 32 * half * p
        T0 = -(-16384 // p)
        Ws = [T0]
        for _ in range(4):
            Ws.append((Ws[-1] + 4 - 5) // 3 + 1)
        for (cin, cout, li, s) in ((128, 512, 2, 3), (512, 1024, 3, 3), (1024, 1024, 4, 1)):
            W = Ws[li]
            Wo = (W + 4 - 5) // s + 1
            x = torch.randn(U, 1, W, cin, device=dev).bfloat16()
            w = (torch.randn(cout, 1, 5, cin, device=dev) * 0.02).bfloat16()
            wt = w.permute(3, 1, 2, 0).contiguous()
            bias = torch.zeros(cout, device=dev)
            tf = timeit(lambda: D.conv2d_fwd(x, w, bias, 1, 5, 1, s, 0, 2, 0.1, True))
            dy = torch.randn(U, 1, Wo, cout, device=dev).bfloat16()
            td = timeit(lambda: D.conv2d_dgrad(dy, wt, 1, W, 1, 5, 1, s, 0, 2, lrelu_y=x, slope=0.1, out_bf16=True))
            fl = 2.0 * U * Wo * 5 * cin * cout
            tot += tf + td
            print(f"U={U:4d} p={p:2d} {cin:4d}->{cout:4d} s{s} M={U*Wo:6d}: fwd {tf:7.1f} us {fl/tf/1e6:6.0f} TF | dgrad {td:7.1f} us {fl/td/1e6:6.0f} TF")
print(f"sum of all timed launches: {tot/1e3:.2f} ms")
for (M, N, Kd) in [(12945, 1024, 5120), (13056, 1024, 2560), (38836, 512, 640), (25600, 1024, 256), (25600, 256, 1024), (16384, 4096, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w = torch.randn(N, Kd, device=dev).bfloat16()
    t = timeit(lambda: K.conv_gemm_bf16(a, w, N, M=M, Trows=M, Tin=M, cin=Kd, out_bf16=True))
    tt = timeit(lambda: torch.matmul(a, w.t()))
    print(f"plain M={M} N={N} K={Kd}: {t:8.1f} us {2.0*M*N*Kd/t/1e6:6.0f} TF | hipBLASLt {tt:8.1f} us {2.0*M*N*Kd/tt/1e6:6.0f} TF")
