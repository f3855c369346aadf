"""Back-to-back timing of the generator-side conv-GEMM shapes (A/B: OSP_GEMM_SMALL=0 vs 1 in separate processes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
SHAPES = [  # (M, T, N, Cin, taps, a32, epi)
    (2048, 2048, 384, 1152, 1, False, "res"), (2048, 2048, 1152, 384, 1, False, "gelu"), (2048, 2048, 1152, 384, 1, False, "gelu_bwd"),
    (2048, 2048, 384, 1152, 1, False, None), (4096, 4096, 256, 1024, 1, False, None), (4096, 4096, 1024, 256, 1, False, "gelu_bwd"),
    (4096, 128, 256, 256, 5, True, "relu"), (4096, 128, 256, 256, 5, True, None), (4096, 128, 384, 384, 3, True, "relu"),
    (2048, 2048, 1088, 384, 1, True, None), (2048, 2048, 256, 1088, 1, True, None), (2048, 2048, 384, 1088, 1, True, None),
    (2048, 64, 384, 256, 7, True, None), (25600, 800, 256, 256, 3, True, None)]
for (M, T, N, Cin, taps, a32, epi) in SHAPES:
    a = torch.randn(M, Cin, device=dev); a = a if a32 else a.bfloat16()
    w = (torch.randn(N, taps, Cin, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    kw = dict(M=M, Trows=T, Tin=T, cin=Cin, taps=taps, a_off=-(taps - 1) // 2)
    if epi == "res":
        res = torch.randn(M, N, device=dev); gamma = torch.randn(N, device=dev); z = torch.empty(M, N, device=dev)
        f = lambda: K.conv_gemm_bf16(a, w, N, epi=K.EPI_SCALE_RES_MASK, bias=bias, gamma=gamma, res=res, aux_out=z, **kw)
    elif epi == "gelu":
        u = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        f = lambda: K.conv_gemm_bf16(a, w, N, epi=K.EPI_GELU, bias=bias, aux_out=u, out_bf16=True, **kw)
    elif epi == "gelu_bwd":
        u = torch.randn(M, N, device=dev).bfloat16()
        f = lambda: K.conv_gemm_bf16(a, w, N, epi=K.EPI_GELU_BWD, aux_in=u, out_bf16=True, **kw)
    elif epi == "relu":
        f = lambda: K.conv_gemm_bf16(a, w, N, epi=K.EPI_RELU, bias=bias, **kw)
    else:
        f = lambda: K.conv_gemm_bf16(a, w, N, **kw)
    t = timeit(f)
    print(f"{os.environ.get('TAG',''):8s} M={M:6d} N={N:5d} K={Cin*taps:5d} taps={taps} a32={int(a32)} {str(epi):9s}: {t:7.1f} us {2.0*M*N*Cin*taps/t/1e6:6.0f} TF")
