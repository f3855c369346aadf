"""osp_conv_gemm_f32 at the index path's / parity modes' shapes; run twice: OSP_GEMM_F32_DMA=1 (direct-to-LDS kernel) and =0 (old kernel)."""
import os, torch
from optispeech_amd import kernels as K
dev = "cuda"
for (M, T, cin, taps, n) in [(8192, 128, 256, 1, 1024), (8192, 128, 1024, 1, 256), (8192, 128, 256, 3, 384), (8192, 128, 384, 3, 384),
                             (25600, 800, 256, 1, 1024), (25600, 800, 1024, 1, 256), (4096, 128, 256, 1, 1024), (4096, 128, 256, 5, 256),
                             (2048, 64, 1152, 1, 384), (2048, 64, 384, 1, 1152), (4096, 128, 1024, 1, 256), (4096, 128, 256, 3, 384), (2048, 64, 384, 1, 1088), (2048, 64, 1088, 1, 256)]:
    x = torch.randn(M, cin, device=dev); w = torch.randn(n, taps, cin, device=dev) * 0.03; b = torch.zeros(n, device=dev)
    fn = lambda: K.conv_gemm(x, w, n, T=T, taps=taps, pad=(taps - 1) // 2, bias=b, epi=K.EPI_GELU)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"S64_BELOW={os.environ.get('OSP_GEMM_F32_S64_BELOW','384')} DMA={os.environ.get('OSP_GEMM_F32_DMA','1')} M={M} Cin={cin} taps={taps} N={n}: {us:.1f} us  {2.0*M*cin*taps*n/us/1e6:.1f} TFLOP/s", flush=True)
