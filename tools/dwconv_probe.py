"""HBM roofline of the fused dwconv7 + LayerNorm forward (A1a) at the shapes of the step."""
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
for (B, T, C) in [(32, 800, 256), (32, 128, 256), (32, 64, 384), (64, 800, 384)]:
    x = torch.randn(B, T, C, device=dev); dw = torch.randn(7, C, device=dev); z = torch.zeros(C, device=dev); o = torch.ones(C, device=dev)
    for save, hb in ((False, False), (True, False), (False, True), (True, True)):
        t = timeit(lambda: K.dwconv7_ln_fwd(x, dw, z, o, z, 1e-6, save, h_bf16=hb))
        byt = B * T * C * (4 + (2 if hb else 4) + (4 if save else 0))
        print(f"FR={os.environ.get('OSP_DWCONV_FR','8')} B={B} T={T} C={C} save={int(save)} h_bf16={int(hb)}: {t:6.1f} us  {byt/t/1e6:6.2f} TB/s = {byt/t/1e6/8*100:4.1f} % of 8 TB/s")
