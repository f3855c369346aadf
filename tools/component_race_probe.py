"""Which component gives run-to-run different results when another process shares the GPU?  Repeats, on fixed inputs and weights:
the three STFT magnitudes (forward + backward), every resolution / period sub-discriminator (scores, input gradient, weight
gradients), and compares each repetition with the first one bit for bit (atomics make weight gradients differ by ~1e-7: a
relative tolerance of 1e-5 separates that from a race).  usage: python tools/component_race_probe.py [f32|bf16] [iters]"""
import sys
import torch

sys.path.insert(0, ".")
from optispeech_amd import precision, spectral                                              # noqa: E402
from optispeech_amd.model.discriminator import MultiPeriodDiscriminator, MultiResolutionDiscriminator   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
precision.set_precision(mode)
torch.manual_seed(3)
dev = "cuda"
B, T = 4, 64 * 256
wav = (torch.randn(B, T, device=dev) * 0.3).clamp_(-1, 1)
mpd = MultiPeriodDiscriminator().to(dev)
mrd = MultiResolutionDiscriminator().to(dev)
first, bad = {}, {}


def check(name, tensors):
    if "sync-each" in sys.argv:                                      # keep the launch queue shallow (diagnostic)
        torch.cuda.synchronize()
    ref = first.setdefault(name, [t.detach().clone() for t in tensors])
    for i, (a, b) in enumerate(zip(tensors, ref)):
        e = ((a - b).norm() / (b.norm() + 1e-30)).item()
        if e > 1e-5:
            bad[name] = bad.get(name, 0) + 1
            if bad[name] <= 3:
                print(f"{name}[{i}] rel {e:.2e}", flush=True)
            break


for it in range(iters):
    for n_fft, hop in ((1024, 256), (2048, 512), (512, 128)):
        x = wav.clone().requires_grad_(True)
        m = spectral.stft_magnitude(x, n_fft, hop, None, None)
        (gx,) = torch.autograd.grad((m * m).sum(), x)
        check(f"stft{n_fft}", [m, gx])
    for n_fft, hop in ((1024, 256), (512, 128)):                    # control: rocFFT through torch.stft, same inputs
        x = wav.clone()
        m = torch.stft(x, n_fft, hop, window=torch.ones(n_fft, device=dev), center=True, pad_mode="reflect", return_complex=True).abs()
        check(f"torch_stft{n_fft}", [m])
    if "stft-only" in sys.argv:
        continue
    for fam, name in ((mrd, "mrd"), (mpd, "mpd")):
        for k, d in enumerate(fam.discriminators):
            x = wav.clone().requires_grad_(True)
            for p in d.parameters():
                p.grad = None
            s, fm = d(x)
            loss = s.float().pow(2).mean() + sum(f.float().abs().mean() for f in fm)
            loss.backward()
            check(f"{name}{k}", [s.float(), x.grad] + [p.grad for p in d.parameters() if p.grad is not None])
torch.cuda.synchronize()
print(f"{mode}: {iters} iterations; deviating repetitions per component: {bad if bad else 'none'}")
