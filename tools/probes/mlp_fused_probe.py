"""Timing of the one-kernel ConvNeXt MLP (csrc/mlp_fused.hip) against the two conv-GEMM launches it replaces, at the synthesise
benchmark's shapes (49k frames; vocoder C=384 I=1152, decoder C=256 I=1024)."""
import sys
import torch
from optispeech_amd import kernels as K, precision

dev = "cuda"
precision.set_precision("bf16")
for (M, C, I) in [(49152, 384, 1152), (49152, 256, 1024), (51200, 384, 1152), (6144, 256, 1024)]:
    g = torch.Generator().manual_seed(1)
    h = torch.randn(M, C, generator=g).to(dev).to(torch.bfloat16)
    x = torch.randn(M, C, generator=g).to(dev)
    W1 = torch.nn.Parameter((torch.randn(I, C, generator=g) / C ** 0.5).to(dev))
    W2 = torch.nn.Parameter((torch.randn(C, I, generator=g) / I ** 0.5).to(dev))
    b1 = torch.zeros(I, device=dev); b2 = torch.zeros(C, device=dev); gamma = torch.ones(C, device=dev)
    mask = torch.ones(M, device=dev)

    def fused():
        return K.convnext_mlp_fused(h, W1, b1, W2, b2, gamma, x, mask)

    def pair():
        gg = K.conv_gemm_bf16(h, K.param_bf16(W1), I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=b1, out_bf16=True)
        return K.conv_gemm_bf16(gg, K.param_bf16(W2), C, M=M, Trows=M, Tin=M, cin=I, epi=K.EPI_SCALE_RES_MASK, bias=b2, gamma=gamma,
                                res=x, rowmask=mask)

    for name, fn in (("fused", fused), ("pair", pair)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        fl = 4.0 * M * C * I
        print(f"M={M} C={C} I={I} {name}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s", flush=True)
    print("  max |fused - pair| =", (fused() - pair()).abs().max().item(), flush=True)
