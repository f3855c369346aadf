"""Diagnostic: generator gradients of the GAN G phase in bf16 mode vs f32 mode, per loss component (small golden model)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision
from tests.test_gpu_training import _small_model, _ref_grads

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "gen_small_gan.npz"), allow_pickle=True)


def run(mode, comp):
    precision.set_precision(mode)
    m = _small_model(g)
    batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    batch.update(sids=None, lids=None)
    m.discriminator.lambda_mel = 0.0
    logs = {}
    for p in m.discriminator.parameters():
        p.requires_grad_(False)
    from optispeech_amd.model.discriminator import _hinge_g, _feature_matching
    out = m._process_batch(batch)
    wav, wav_hat = out["wav"], out["wav_hat"]
    d = m.discriminator
    if comp == "all":
        loss = d.forward_gen(wav, wav_hat)[0]
    elif comp in ("gen_mp", "fm_mp"):
        _, g_mp, fr, fg = d.multiperioddisc(y=wav, y_hat=wav_hat)
        loss = _hinge_g(g_mp) if comp == "gen_mp" else _feature_matching(fr, fg)
    elif comp in ("gen_mrd", "fm_mrd"):
        _, g_mr, fr, fg = d.multiresddisc(y=wav, y_hat=wav_hat)
        loss = _hinge_g(g_mr) if comp == "gen_mrd" else _feature_matching(fr, fg)
    elif comp == "stft":
        loss = d._get_mr_stft_loss(wav, wav_hat)
    print(comp, mode, "loss", loss.item())
    loss.backward()
    return {k: v.double().cpu() for k, v in _ref_grads(m.generator).items() if k.startswith("vocoder.")}, logs


for comp in sys.argv[1:] or ["all"]:
    a, la = run("f32", comp)
    b, lb = run("bf16", comp)
    print("==", comp)
    for k in sorted(a):
        na, nb = a[k].norm().item(), b[k].norm().item()
        if na < 1e-9:
            continue
        cos = torch.nn.functional.cosine_similarity(a[k].flatten(), b[k].flatten(), dim=0).item()
        print(f"{k:50s} f32 {na:10.4g} bf16 {nb:10.4g} ratio {nb/na:6.3f} cos {cos:6.3f}")
