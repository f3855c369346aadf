#!/usr/bin/env python3
"""Decoder pwconv1 shape (M=25600, 256 -> 1024): what costs the time -- f32 A operand, GELU epilogue, output type? (diagnostic)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K, precision
precision.set_precision("bf16")
dev = "cuda"
M, C, I = 25600, 256, 1024
h32 = torch.randn(M, C, device=dev); h16 = h32.bfloat16()
w = (torch.randn(I, C, device=dev) * 0.05).bfloat16(); b = torch.randn(I, device=dev)
g16 = torch.randn(M, I, device=dev).bfloat16(); w2 = (torch.randn(C, I, device=dev) * 0.05).bfloat16()
x = torch.randn(M, C, device=dev); gam = torch.ones(C, device=dev); b2 = torch.randn(C, device=dev)


def t(name, f, flop, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / reps * 1e3
    print(f"{name:58s} {us:7.1f} us  {flop / us / 1e6:6.0f} TF")


fl = 2.0 * M * C * I
t("pwconv1 A f32,  GELU, out bf16 (decoder today)", lambda: K.conv_gemm_bf16(h32, w, I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=b, out_bf16=True), fl)
t("pwconv1 A bf16, GELU, out bf16", lambda: K.conv_gemm_bf16(h16, w, I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=b, out_bf16=True), fl)
t("pwconv1 A bf16, none, out bf16", lambda: K.conv_gemm_bf16(h16, w, I, M=M, Trows=M, Tin=M, cin=C, bias=b, out_bf16=True), fl)
t("pwconv1 A f32,  none, out bf16", lambda: K.conv_gemm_bf16(h32, w, I, M=M, Trows=M, Tin=M, cin=C, bias=b, out_bf16=True), fl)
t("pwconv1 A bf16, GELU, out f32", lambda: K.conv_gemm_bf16(h16, w, I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=b), fl)
t("pwconv2 A bf16, SCALE_RES_MASK, out f32 (decoder today)", lambda: K.conv_gemm_bf16(g16, w2, C, M=M, Trows=M, Tin=M, cin=I, epi=K.EPI_SCALE_RES_MASK, bias=b2, gamma=gam, res=x), fl)
t("pwconv2 A bf16, none, out f32", lambda: K.conv_gemm_bf16(g16, w2, C, M=M, Trows=M, Tin=M, cin=I, bias=b2), fl)
t("cast f32 -> bf16 of h (M x 256)", lambda: K.cast_bf16(h32), 0.0)
