#!/usr/bin/env python3
"""Timing of the Cin = 1 first-layer kernels (csrc/smallcin.hip) on the bench shapes (diagnostic).
OSP_SMALLCIN_VALU=1 selects the VALU kernels for comparison."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
shapes = [("MRD 1024", 64, 65, 513, 64, 5, 7, 2, 2, 2, 3), ("MRD 2048", 64, 33, 1025, 64, 5, 7, 2, 2, 2, 3),
          ("MRD 512", 64, 129, 257, 64, 5, 7, 2, 2, 2, 3), ("MPD p=2", 128, 1, 8192, 32, 1, 5, 1, 3, 0, 2)]
for name, U, Hin, Win, cout, KH, KW, sh, sw, ph, pw in shapes:
    if name.startswith("MRD"):
        KH, KW, sh, sw, ph, pw = 7, 5, 2, 2, 3, 2
    x = torch.randn(U, Hin, Win, device=dev)
    w = torch.randn(cout, KH * KW, device=dev) * 0.1
    b = torch.zeros(cout, device=dev)
    Ho, Wo = (Hin + 2 * ph - KH) // sh + 1, (Win + 2 * pw - KW) // sw + 1
    kw = dict(U=U, Hin=Hin, Win=Win, Ho=Ho, Wo=Wo, cout=cout, KH=KH, KW=KW, sh=sh, sw=sw, ph=ph, pw=pw)
    y = K.smallcin_fwd(x, w, b, slope=0.1, out_bf16=True, **kw)
    dy = torch.randn_like(y)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    tf = t(lambda: K.smallcin_fwd(x, w, b, slope=0.1, out_bf16=True, **kw))
    tw = t(lambda: K.smallcin_wgrad(x, dy, dw, db, **kw))
    mb = y.numel() * 2 / 1e6
    print(f"{name}: rows {y.shape[0]} fwd {tf:7.1f} us ({mb/tf*1e3:6.0f} GB/s of y)  wgrad {tw:7.1f} us ({mb/tw*1e3:6.0f} GB/s of dy)")
