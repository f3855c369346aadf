import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import kernels as K
dev = "cuda"
def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (M, N, Kd) in [(13056, 1024, 5120), (13056, 1024, 2560), (16384, 4096, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, Kd, device=dev).bfloat16(); w = torch.randn(N, Kd, device=dev).bfloat16()
    t = timeit(lambda: K.conv_gemm_bf16(a, w, N, M=M, Trows=M, Tin=M, cin=Kd, out_bf16=True))
    print(f"{os.environ.get('TAG','')} M={M} N={N} K={Kd}: {t:8.1f} us {2.0*M*N*Kd/t/1e6:6.0f} TF")
