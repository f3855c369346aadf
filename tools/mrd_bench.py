#!/usr/bin/env python3
"""Per-layer timing of the DiscriminatorR stack on the bf16 conv-GEMM (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
from optispeech_amd import disc_ops as D
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
B = 32
tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}
for (n_fft, hop) in ((1024, 256), (2048, 512), (512, 128)):
    H, W = 1 + 16384 // hop, n_fft // 2 + 1
    x = torch.rand(B, H, W, 1, device=dev)
    print(f"res {n_fft}: H={H} W={W}")
    cin = 1
    for i, sp in enumerate(D.MRD_SPEC):
        KH, KW, sh, sw, ph, pw = sp
        cout = 64 if i < 5 else 1
        w = K.cast_bf16(torch.randn(cout, KH, KW, cin, device=dev) * 0.05)
        bias = torch.zeros(cout, device=dev)
        y = D.conv2d_fwd(x, w, bias, *sp, 0.1 if i < 5 else None, i < 5)
        fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * KH * KW * cin * cout
        tf = t(lambda: D.conv2d_fwd(x, w, bias, *sp, 0.1 if i < 5 else None, i < 5))
        dy = torch.randn_like(y)
        wt = D.transpose_weight2d(w)
        td = t(lambda: D.conv2d_dgrad(dy, wt, x.shape[1], x.shape[2], *sp, lrelu_y=x if i > 0 else None, out_bf16=i > 0))
        tw = t(lambda: D.conv2d_wgrad(dy, x, *sp))
        print(f"  L{i} {cin:3d}->{cout:3d} rows {y.shape[0]*y.shape[1]*y.shape[2]:8d} {fl/1e9:6.2f} GF fwd {tf:7.3f} dgrad {td:7.3f} wgrad {tw:7.3f} ms")
        tot["fwd"] += tf; tot["dgrad"] += td; tot["wgrad"] += tw
        x, cin = y, cout
print(tot)
