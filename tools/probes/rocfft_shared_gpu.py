"""Repo-code-free reproducer: only torch (rocFFT behind torch.stft / torch.fft.rfft, rocBLAS behind matmul).  Run one instance
alone, then two side by side on ONE MI355X; each repetition recomputes from the unchanged input and is compared with the first
repetition.  usage: python tools/probes/rocfft_shared_gpu.py [iters]      (never imports optispeech_amd)"""
import os
import sys

import torch

assert "optispeech_amd" not in sys.modules
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.manual_seed(3)
dev = "cuda"
B, T = 8, 64 * 256 * 4
wav = (torch.randn(B, T, device=dev) * 0.3).clamp_(-1, 1)
a, b = torch.randn(2048, 2048, device=dev), torch.randn(2048, 2048, device=dev)
first, bad, worst = {}, {}, {}


def check(name, t):
    ref = first.setdefault(name, t.detach().clone())
    if not torch.equal(t, ref):
        e = ((t - ref).abs().max() / ref.abs().max()).item()
        bad[name] = bad.get(name, 0) + 1
        worst[name] = max(worst.get(name, 0.0), e)


for it in range(iters):
    for n_fft, hop in ((1024, 256), (2048, 512), (512, 128)):
        m = torch.stft(wav, n_fft, hop, window=torch.ones(n_fft, device=dev), center=True, pad_mode="reflect", return_complex=True)
        check(f"torch.stft{n_fft}", torch.view_as_real(m))
    check("torch.fft.rfft(16384)", torch.view_as_real(torch.fft.rfft(wav.view(-1, 16384))))
    check("matmul2048", a @ b)
    check("cumsum", torch.cumsum(wav, 1))
torch.cuda.synchronize()
print(f"pid {os.getpid()}: {iters} repetitions; deviating repetitions {bad if bad else 'none'}; worst max-rel deviation "
      f"{ {k: f'{v:.1e}' for k, v in worst.items()} }")
