"""Diagnostic: where the MR-STFT-loss gradient diverges between bf16 and f32 modes (small golden model)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision
from tests.test_gpu_training import _small_model, _ref_grads

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "gen_small_gan.npz"), allow_pickle=True)
cosf = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


def run(mode, dwav_in=None):
    precision.set_precision(mode)
    m = _small_model(g)
    batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    batch.update(sids=None, lids=None)
    out = m._process_batch(batch)
    wav, wav_hat = out["wav"], out["wav_hat"]
    wav_hat.retain_grad()
    d = m.discriminator
    if dwav_in is None:
        loss = d._get_mr_stft_loss(wav, wav_hat)
        loss.backward()
    else:
        wav_hat.backward(dwav_in)
    return wav.detach(), wav_hat.detach(), wav_hat.grad.clone(), {k: v.double().cpu() for k, v in _ref_grads(m.generator).items() if k.startswith("vocoder.")}


wa, ha, da, ga = run("f32")
wb, hb, db, gb = run("bf16")
print("wav_hat shape", ha.shape, "rms", ha.pow(2).mean().sqrt().item(), "diff rms", (ha - hb).pow(2).mean().sqrt().item())
print("dwav f32 norm", da.norm().item(), "bf16 norm", db.norm().item(), "cos", cosf(da, db))
# same dwav through both backward paths
_, _, _, ga2 = run("f32", da)
_, _, _, gb2 = run("bf16", da)
for k in ("vocoder.head.linear_2.weight", "vocoder.head.linear_1.weight", "vocoder.backbone.convnext.0.pwconv1.weight", "vocoder.embed.weight"):
    print(k, "same-dwav: f32", ga2[k].norm().item(), "bf16", gb2[k].norm().item(), "cos", cosf(ga2[k], gb2[k]),
          "| own-dwav cos", cosf(ga[k], gb[k]))
# stft loss gradient on f32 wav_hat + tiny noise
from optispeech_amd.model import discriminator as D
precision.set_precision("f32")
m = _small_model(g)
for eps in (1e-4, 1e-3, 3e-3):
    h2 = (ha * (1 + eps * torch.randn_like(ha))).requires_grad_(True)
    m.discriminator._get_mr_stft_loss(wa, h2).backward()
    print("rel noise", eps, "dwav cos", cosf(h2.grad, da), "norm", h2.grad.norm().item())
