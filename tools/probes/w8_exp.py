#!/usr/bin/env python3
"""8-wave conv-GEMM at the DiscriminatorP layer-4 / layer-5 shapes and at tile counts that fill 0.4 - 2.0 waves of the chip."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import kernels as K
dev = "cuda"


def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (U, T, cin, n, st) in [(128, 102, 1024, 1024, 1), (64, 102, 1024, 1024, 1), (96, 102, 1024, 1024, 1), (160, 102, 1024, 1024, 1), (256, 102, 1024, 1024, 1),
                           (128, 304, 512, 1024, 3), (704, 19, 1024, 1024, 1)]:
    Tout = (T + 4 - 5) // st + 1
    M = U * Tout
    a = torch.randn(U * T, cin, device=dev).bfloat16(); w = torch.randn(n, 5, cin, device=dev).bfloat16()
    t = timeit(lambda: K.conv_gemm_bf16(a, w, n, M=M, Trows=Tout, Tin=T, cin=cin, taps=5, a_step=st, a_off=-2, out_bf16=True))
    tiles = -(-M // 256) * (n // 256)
    if os.environ.get("CHECK", "1") == "1":
        import torch.nn.functional as F
        out = K.conv_gemm_bf16(a, w, n, M=M, Trows=Tout, Tin=T, cin=cin, taps=5, a_step=st, a_off=-2, out_bf16=False)
        ref = F.conv1d(a.float().view(U, T, cin).transpose(1, 2), w.float().permute(0, 2, 1).contiguous(), stride=st, padding=2).transpose(1, 2).reshape(M, n)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        print(f"         max |err| / max |ref| = {err:.2e}", flush=True)
        assert err < 2e-3
    print(f"{os.environ.get('TAG', ''):8s} U={U} T={T} {n}<-{cin} s{st} M={M} tiles={tiles}: {t:8.1f} us {2.0 * M * n * cin * 5 / t / 1e6:6.0f} TF", flush=True)

# the vendor library on the same amount of work (a plain GEMM with K = 5 * Cin), for the practical ceiling of the regime
for M in (6528, 9792, 13056, 16320, 26112):
    a = torch.randn(M, 5120, device=dev).bfloat16(); w = torch.randn(1024, 5120, device=dev).bfloat16()
    t = timeit(lambda: torch.matmul(a, w.t()))
    print(f"{os.environ.get('TAG', ''):8s} hipBLASLt (torch.matmul) M={M} 1024<-5120: {t:8.1f} us {2.0 * M * 1024 * 5120 / t / 1e6:6.0f} TF", flush=True)
