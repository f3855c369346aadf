"""osp_conv_gemm_f32 (exact) against osp_conv_gemm_f32_split (f32 operands, three bf16 MFMAs per product) at the "mixed" parity mode's
generator shapes: microseconds per launch, TFLOP/s and the max error of each against float64 relative to the result's scale."""
import torch
import torch.nn.functional as F
from optispeech_amd import kernels as K, precision
dev = "cuda"
SH = [(8192, 128, 256, 1, 1024), (8192, 128, 1024, 1, 256), (25600, 800, 256, 1, 1024), (25600, 800, 1024, 1, 256), (25600, 800, 256, 3, 256),
      (4096, 128, 256, 1, 1024), (4096, 128, 256, 5, 256), (2048, 64, 1152, 1, 384), (2048, 64, 384, 1, 1152), (4096, 128, 1024, 1, 256),
      (2048, 64, 384, 1, 1088), (2048, 64, 1088, 1, 256)]
for (M, T, cin, taps, n) in SH:
    x = torch.randn(M, cin, device=dev); w = torch.randn(n, taps, cin, device=dev) * 0.03; b = torch.zeros(n, device=dev)
    want = None
    if taps == 1:
        want = F.linear(x.double(), w.view(n, cin).double())
    line = f"M={M} Cin={cin} taps={taps} N={n}:"
    for mode in ("f32", "mixed"):
        precision.set_precision(mode)
        fn = lambda: K.conv_gemm(x, w, n, T=T, taps=taps, pad=(taps - 1) // 2, bias=b)
        for _ in range(3): y = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        err = ((y.double() - want).abs().max() / want.abs().max()).item() if want is not None else float("nan")
        line += f"  {'exact' if mode == 'f32' else 'split'} {us:7.1f} us {2.0*M*cin*taps*n/us/1e6:6.1f} TFLOP/s err {err:.1e}"
    print(line, flush=True)
precision.set_precision("f32")
