#!/usr/bin/env python3
"""A/B of the 64 <- 64 DiscriminatorR layers: conv2d_panel_n64_kernel (OSP_N64_PANEL=1) vs the per-tap glds n64 kernel (default).
Run twice:  OSP_N64_PANEL=1 python tools/probes/panel_probe.py ; python tools/probes/panel_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import kernels as K, disc_ops as D
dev = "cuda"
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
U = int(os.environ.get("U", "64"))
print("OSP_N64_PANEL =", os.environ.get("OSP_N64_PANEL", "0"), " U =", U)
tot = [0.0, 0.0]
ONLY = os.environ.get("ONLY", "0") == "1"              # counters: the first resolution's layer 1 only
for (n_fft, hop) in ((1024, 256), (2048, 512), (512, 128))[:1 if ONLY else 3]:
    H, W = 1 + 8192 // hop, n_fft // 2 + 1
    H, W = (H + 4 - 5) // 2 + 1, (W + 6 - 7) // 2 + 1                       # layer 0 output
    for i, sp in enumerate(D.MRD_SPEC[1:2 if ONLY else 5]):
        KH, KW, sh, sw, ph, pw = sp
        x = torch.randn(U, H, W, 64, device=dev).to(torch.bfloat16)
        w = K.cast_bf16(torch.randn(64, KH, KW, 64, device=dev) * 0.05)
        bias = torch.zeros(64, device=dev)
        y = D.conv2d_fwd(x, w, bias, *sp, 0.1, True)
        fl = 2.0 * y.numel() * KH * KW * 64
        tf = t(lambda: D.conv2d_fwd(x, w, bias, *sp, 0.1, True))
        dy = torch.randn_like(y)
        wt = D.transpose_weight2d(w)
        td = t(lambda: D.conv2d_dgrad(dy, wt, H, W, *sp, lrelu_y=x, out_bf16=True))
        print(f"res {n_fft:4d} L{i+1} in {H:3d}x{W:3d} rows {y.shape[0]*y.shape[1]*y.shape[2]:7d} {fl/1e9:6.2f} GF  fwd {tf:7.1f} us {fl/tf/1e6:6.0f} TF/s   dgrad {td:7.1f} us {fl/td/1e6:6.0f} TF/s")
        tot[0] += tf; tot[1] += td
        H, W = y.shape[1], y.shape[2]
print(f"total fwd {tot[0]:.1f} us  dgrad {tot[1]:.1f} us")
