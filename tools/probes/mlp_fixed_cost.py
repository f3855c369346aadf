"""Fixed vs per-chunk time of the one-kernel MLP for a workgroup that has its CU to itself (48 workgroups): hidden width sweep."""
import torch
from optispeech_amd import kernels as K, precision
precision.set_precision("bf16")
dev = "cuda"
for C in (256, 384):
    for variant in ("1w",):
        res = []
        for I in (128, 256, 512, 1024, 2048):
            M = 6144
            g = torch.Generator().manual_seed(1)
            h = torch.randn(M, C, generator=g).to(dev).to(torch.bfloat16); x = torch.randn(M, C, generator=g).to(dev)
            W1 = torch.nn.Parameter((torch.randn(I, C, generator=g) / C ** 0.5).to(dev)); W2 = torch.nn.Parameter((torch.randn(C, I, generator=g) / I ** 0.5).to(dev))
            b1 = torch.zeros(I, device=dev); b2 = torch.zeros(C, device=dev); gamma = torch.ones(C, device=dev)
            fn = lambda: K.convnext_mlp_fused(h, W1, b1, W2, b2, gamma, x, None, None)
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res.append((I, e0.elapsed_time(e1) * 50))
        per = (res[-1][1] - res[0][1]) / ((res[-1][0] - res[0][0]) / 128)
        print(f"C={C} {variant}: " + "  ".join(f"I={i}: {t:.1f} us" for i, t in res) + f"   -> {per:.2f} us per 128 hidden units, fixed ~{res[0][1] - per:.1f} us", flush=True)
