"""Fused LayerNorm + depthwise-conv backward vs the two-kernel path at the decoder shape (32 x 800 x 256), back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
def timeit(f, reps=100):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (B, T, C) in [(32, 800, 256), (32, 128, 256)]:
    M = B * T
    dh, xhat = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
    rstd = torch.rand(M, device=dev) + 0.5
    lnw, x, dw = torch.randn(C, device=dev), torch.randn(B, T, C, device=dev), torch.randn(7, C, device=dev)
    dres, rm = torch.randn(B, T, C, device=dev), torch.ones(M, device=dev)
    g = [torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(7, C, device=dev), torch.zeros(C, device=dev)]
    def pair():
        dc = K.layernorm_bwd(dh, xhat, None, rstd, lnw, g[0], g[1])
        return K.dwconv7_bwd(dc.view(B, T, C), x, dw, dres, rm, g[2], g[3])
    def fused():
        return K.ln_dwconv7_bwd(dh, xhat, rstd, lnw, x, dw, dres, rm, g[0], g[1], g[2], g[3])
    def fused_nograd():
        return K.ln_dwconv7_bwd(dh, xhat, rstd, lnw, x, dw, dres, rm, None, None, None, None)
    def fwd():
        return K.dwconv7_ln_fwd(x, dw, g[3], lnw, g[1], 1e-6, True, h_bf16=True)
    print(f"   fused without parameter gradients {timeit(fused_nograd):6.1f} us; forward (bf16 h, xhat saved) {timeit(fwd):6.1f} us")
    print(f"   stand-alone layernorm_bwd {timeit(lambda: K.layernorm_bwd(dh, xhat, None, rstd, lnw, g[0], g[1])):6.1f} us")
    tp, tf = timeit(pair), timeit(fused)
    alg = 5 * M * C * 4
    print(f"{os.environ.get('TAG','')} B={B} T={T} C={C}: pair {tp:6.1f} us, fused {tf:6.1f} us = {alg/tf/1e3:6.0f} GB/s ({alg/tf/1e3/8000*100:4.1f} % of 8 TB/s; algorithmic {alg/1e6:.1f} MB)")
