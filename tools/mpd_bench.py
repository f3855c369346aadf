#!/usr/bin/env python3
"""Per-layer timing of the DiscriminatorP stack on the bf16 conv-GEMM (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
from optispeech_amd.disc_ops import conv1d_strided_fwd, conv1d_strided_dgrad, transpose_weight

dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

B = int(os.environ.get("NB", "32"))
tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}
for period in (2, 3, 5, 7, 11):
    T0 = (16384 + period - 1) // period
    U = B * period
    ch = [1, 32, 128, 512, 1024, 1024]
    strides = [3, 3, 3, 3, 1]
    x = torch.randn(U, T0, 1, device=dev)
    Tin = T0
    print(f"period {period}: U={U} T0={T0}")
    for i in range(5):
        cin, cout, st = ch[i], ch[i + 1], strides[i]
        w = torch.randn(cout, 5, cin, device=dev) * 0.05
        wb = K.cast_bf16(w)
        bias = torch.zeros(cout, device=dev)
        Tout = (Tin + 4 - 5) // st + 1
        y = conv1d_strided_fwd(x, wb, bias, 5, st, 2, 0.1, True)
        fl = 2.0 * U * Tout * 5 * cin * cout
        tf = t(lambda: conv1d_strided_fwd(x, wb, bias, 5, st, 2, 0.1, True))
        dy = torch.randn_like(y)
        wt = transpose_weight(wb)
        td = t(lambda: conv1d_strided_dgrad(dy, wb, Tin, cin, 5, st, 2, lrelu_y=x if i > 0 else None, out_bf16=i > 0, wt=wt))
        dw = torch.zeros(cout, 5, cin, device=dev); db = torch.zeros(cout, device=dev)
        tw = t(lambda: K.conv_wgrad_bf16(dy.view(U * Tout, cout), x.view(U * Tin, cin), dw, db, M=U * Tout, Trows=Tout, Tin=Tin, n=cout, cin=cin, taps=5, pad=2, x_step=st))
        print(f"  L{i} {cin:5d}->{cout:5d} s{st} rows {U*Tout:8d} {fl/1e9:7.1f} GF  fwd {tf:7.3f} ms ({fl/tf/1e9:6.1f} TF)  dgrad {td:7.3f} ms ({fl/td/1e9:6.1f} TF)  wgrad {tw:7.3f} ms ({fl/tw/1e9:6.1f} TF)")
        tot["fwd"] += tf; tot["dgrad"] += td; tot["wgrad"] += tw
        x, Tin = y, Tout
print(tot)
