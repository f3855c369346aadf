#!/usr/bin/env python3
"""Time osp_conv2d_dgrad_bf16 on the DiscriminatorP shapes at half / full batch, with and without the fused epilogue inputs
(diagnostic: the generator-phase launches (half batch) were as slow as the discriminator-phase ones)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import disc_ops as D, precision
precision.set_precision("bf16")
dev = "cuda"


def run(U, W, Cin, Cout, sw, lrelu, extra, extra_bf16=True, reps=20):
    KH, KW, ph, pw = 1, 5, 0, 2
    Wo = (W + 2 * pw - KW) // sw + 1
    dy = torch.randn(U, 1, Wo, Cout, device=dev).bfloat16()
    wt = torch.randn(Cin, KH, KW, Cout, device=dev).bfloat16() * 0.01
    y = torch.randn(U, 1, W, Cin, device=dev).bfloat16() if lrelu else None
    e = (torch.randn(U, 1, W, Cin, device=dev).bfloat16() if extra_bf16 else torch.randn(U, 1, W, Cin, device=dev)) if extra else None
    f = lambda: D.conv2d_dgrad(dy, wt, 1, W, KH, KW, 1, sw, ph, pw, lrelu_y=y, extra=e, out_bf16=True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    tf = 2.0 * U * W * Cin * Cout * KW / sw / (us * 1e-6) / 1e12
    print(f"U={U:4d} W={W:4d} {Cout}->{Cin} s{sw} lrelu={int(lrelu)} extra={int(extra)}({'bf16' if extra_bf16 else 'f32'}): {us:7.1f} us {tf:6.0f} TF")


for U in (352, 704):
    for lrelu, extra, eb in ((False, False, True), (True, False, True), (True, True, True), (True, True, False)):
        run(U, 56, 512, 1024, 3, lrelu, extra, eb)
for U in (352, 704):
    for lrelu, extra, eb in ((False, False, True), (True, True, True), (True, True, False)):
        run(U, 19, 1024, 1024, 1, lrelu, extra, eb)
for U in (352, 704):
    for lrelu, extra, eb in ((False, False, True), (True, True, True), (True, True, False)):
        run(U, 166, 128, 512, 3, lrelu, extra, eb)
