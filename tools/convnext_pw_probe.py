"""The four pointwise GEMMs of a decoder ConvNeXt block (B=32, T=800, C=256, I=1024) in the performance mode, 50 back-to-back
launches each (HIP events): time, algorithmic bytes (each operand once) and the HBM fraction.  python tools/convnext_pw_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (B, T, C, I) in ((32, 800, 256, 1024), (32, 64, 384, 1152), (32, 128, 256, 1024)):
    M = B * T
    g = torch.Generator().manual_seed(0)
    h = torch.randn(M, C, generator=g).to(dev).bfloat16()
    x = torch.randn(M, C, generator=g).to(dev)
    W1 = (torch.randn(I, C, generator=g) * 0.05).to(dev).bfloat16(); b1 = torch.zeros(I, device=dev)
    W2 = (torch.randn(C, I, generator=g) * 0.05).to(dev).bfloat16(); b2 = torch.zeros(C, device=dev)
    W2t = W2.t().contiguous(); W1t = W1.t().contiguous()
    gamma = torch.full((C,), 0.25, device=dev)
    rowmask = torch.ones(M, device=dev); rowscale = torch.ones(M, device=dev)
    u = torch.empty(M, I, device=dev, dtype=torch.bfloat16); z = torch.empty(M, C, device=dev)
    gg = torch.empty(M, I, device=dev, dtype=torch.bfloat16); y = torch.empty(M, C, device=dev)
    dys = torch.randn(M, C, generator=g).to(dev).bfloat16(); du = torch.empty(M, I, device=dev, dtype=torch.bfloat16); dh = torch.empty(M, C, device=dev)
    print(f"B={B} T={T} C={C} I={I}")
    rows = [
        ("pw1 plain (bf16 out)", lambda: K.conv_gemm_bf16(h, W1, I, M=M, Trows=M, Tin=M, cin=C, out=gg, out_bf16=True), M * C * 2 + I * C * 2 + M * I * 2),
        ("pw1 GELU no aux", lambda: K.conv_gemm_bf16(h, W1, I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=b1, out=gg, out_bf16=True), M * C * 2 + I * C * 2 + M * I * 2),
        ("pw1 GELU + u", lambda: K.conv_gemm_bf16(h, W1, I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=b1, aux_out=u, out=gg, out_bf16=True), M * C * 2 + I * C * 2 + 2 * M * I * 2),
        ("pw2 plain (f32 out)", lambda: K.conv_gemm_bf16(gg, W2, C, M=M, Trows=M, Tin=M, cin=I, out=y), M * I * 2 + I * C * 2 + M * C * 4),
        ("pw2 scale/res/mask + z", lambda: K.conv_gemm_bf16(gg, W2, C, M=M, Trows=M, Tin=M, cin=I, epi=K.EPI_SCALE_RES_MASK, bias=b2, gamma=gamma, res=x, rowmask=rowmask, rowscale=rowscale, aux_out=z, out=y), M * I * 2 + I * C * 2 + 3 * M * C * 4),
        ("du GELU' (bf16 out)", lambda: K.conv_gemm_bf16(dys, W2t, I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU_BWD, aux_in=u, out=du, out_bf16=True), M * C * 2 + I * C * 2 + 2 * M * I * 2),
        ("dh plain (f32 out)", lambda: K.conv_gemm_bf16(du, W1t, C, M=M, Trows=M, Tin=M, cin=I, out=dh), M * I * 2 + I * C * 2 + M * C * 4),
    ]
    for name, fn, by in rows:
        us = t(fn)
        print(f"  {name:28s} {us:7.1f} us  {by / 1e6:6.1f} MB  {by / us / 1e6:6.2f} TB/s = {by / us / 1e6 / 8 * 100:4.1f} % of 8 TB/s")
