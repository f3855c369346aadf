#!/usr/bin/env python3
"""us per launch and TB/s of the feature-matching kernels (osp_l1_sum_multi / osp_l1_sign_multi) on a family-sized list of bf16 maps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import kernels as K
dev = "cuda"
sizes = [32 * 1024 * 1365, 32 * 512 * 455 * 4, 32 * 128 * 4095, 32 * 64 * 513 * 50, 32 * 64 * 257 * 50]      # ~0.2 G elements
a = [torch.randn(n, device=dev).to(torch.bfloat16) for n in sizes]
b = [torch.randn(n, device=dev).to(torch.bfloat16) for n in sizes]
g = torch.tensor([0.5], device=dev)
out = torch.zeros((), device=dev)
tot = sum(sizes)
for name, fn, nb in (("l1_sum_multi ", lambda: K.l1_sum_multi(a, b, out), 4 * tot), ("l1_sign_multi", lambda: K.l1_sign_multi(a, b, g), 6 * tot)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name}: {us:7.1f} us per launch over {tot / 1e6:.0f} M bf16 elements = {nb / us / 1e6:.2f} TB/s")
