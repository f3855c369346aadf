"""GPU diagnostic: how far the bf16 (bench) precision mode is from the f32 mode / the reference goldens.  Prints the numbers
the tolerances of tests/test_gpu_bf16.py are set from."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import schema as S
from optispeech_amd import precision
from optispeech_amd.config import ModelConfig, make_generator
from tests.test_gpu_training import _small_model, _ref_grads
from tests.test_gpu_generator import _small_cfg, _ref_grad

G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
rel = lambda a, b: ((a.double().cpu() - torch.as_tensor(np.asarray(b)).double()).abs().max() / torch.as_tensor(np.asarray(b)).double().abs().max()).item()

for name, cfg, scfg in (("gen_small_am", _small_cfg(), S.SMALL), ("gen_full_b2", ModelConfig().no_dropout(), S.Cfg())):
    g = G(name)
    res = {}
    for mode in ("f32", "bf16"):
        precision.set_precision(mode)
        gen = make_generator(cfg).to("cuda").train()
        W = S.make_weights(S.generator_schema(scfg), int(g["seed"]))
        gen.load_state_dict({k[len("generator."):]: v for k, v in W.items()})
        gen.segment_rand01 = torch.from_numpy(g["rand01"])
        b = {k[3:]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith("in_") and k != "in_wav"}
        out = gen(b["x"], b["x_lengths"], b["mel"], b["mel_lengths"], b["pitches"], b["energies"], None, None)
        out["loss"].backward()
        torch.cuda.synchronize()
        gn = {k: (_ref_grad(gen, k).double().norm().item() if _ref_grad(gen, k) is not None else None) for k in g["grad_g_names"].tolist()}
        res[mode] = (out, gn)
    o32, o16 = res["f32"][0], res["bf16"][0]
    print(f"== {name}: durations equal f32/golden {np.array_equal(o32['_aux']['durations'].cpu().numpy(), g['durations'])}, "
          f"bf16/golden {np.array_equal(o16['_aux']['durations'].cpu().numpy(), g['durations'])} "
          f"(bf16 differing tokens {(o16['_aux']['durations'].cpu().numpy() != g['durations']).sum()})")
    print("   start_idx equal bf16:", np.array_equal(o16["start_idx"].cpu().numpy(), g["start_idx"]))
    for k in ("loss", "align_loss", "duration_loss", "pitch_loss", "energy_loss"):
        print(f"   {k}: golden {float(g[k]):.6f} f32 {o32[k].item():.6f} bf16 {o16[k].item():.6f} rel {abs(o16[k].item()-float(g[k]))/abs(float(g[k])):.2e}")
    w32, w16 = o32["wav_hat"].detach(), o16["wav_hat"].detach()
    print(f"   wav_hat: bf16 vs f32 max-rel {((w16-w32).abs().max()/w32.abs().max()).item():.3e}  rms-rel {((w16-w32).norm()/w32.norm()).item():.3e}")
    if "wav_hat" in g.files:
        print(f"   wav_hat vs golden: f32 {rel(w32, g['wav_hat']):.3e} bf16 {rel(w16, g['wav_hat']):.3e}")
    else:
        print(f"   wav_hat l2 golden {float(g['wav_hat_l2']):.6f} f32 {w32.double().norm().item():.6f} bf16 {w16.double().norm().item():.6f}")
    for k in ("log_p_attn", "dec", "y_up", "enc"):
        pass
    worst = []
    for k, n in zip(g["grad_g_names"].tolist(), g["grad_g_norms"].tolist()):
        a, c = res["f32"][1][k], res["bf16"][1][k]
        if n > 1e-6:
            worst.append((abs(c - n) / n, abs(a - n) / n, k))
    worst.sort(reverse=True)
    print("   AM/vocoder grad-norm deviation vs golden (bf16, f32, name), worst 8:")
    for w in worst[:8]:
        print(f"      {w[0]:.3e} {w[1]:.3e} {w[2]}")
    am = [w for w in worst if not w[2].startswith("vocoder.")]
    print(f"   worst non-vocoder: {am[0] if am else None}")

# GAN step small: D grads
g = G("gen_small_gan")
for mode in ("f32", "bf16"):
    precision.set_precision(mode)
    m = _small_model(g)
    batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    batch.update(sids=None, lids=None)
    m.discriminator.lambda_mel = 0.0
    logs = {}
    for p in m.discriminator.parameters():
        p.requires_grad_(False)
    loss_g, (wav, wav_hat) = m.training_step_g(batch, True, logs)
    print(f"== gen_small_gan [{mode}] wav_hat vs golden {rel(wav_hat.detach(), g['wav_hat']):.3e}  loss_g {loss_g.item():.5f} vs {float(g['loss_g']):.5f}")
    loss_g.backward()
    gg = _ref_grads(m.generator)
    dev = sorted(((abs(gg[k].double().norm().item() - n) / n, k) for k, n in zip(g["grad_g_names"].tolist(), g["grad_g_norms"].tolist()) if n > 1e-6 and gg[k] is not None), reverse=True)
    print("   G grad-norm dev worst 6:", [(f"{a:.2e}", k) for a, k in dev[:6]])
    print("   G grad-norm dev worst non-vocoder:", [(f"{a:.2e}", k) for a, k in dev if not k.startswith("vocoder.")][:3])
    for p in m.discriminator.parameters():
        p.requires_grad_(True)
    m.optimizers()[1].zero_grad()
    loss_d = m.training_step_d(batch, (wav, wav_hat.detach()), logs)
    loss_d.backward()
    gd = _ref_grads(m.discriminator)
    dev = sorted(((abs(gd[k].double().norm().item() - n) / n, n, k) for k, n in zip(g["grad_d_names"].tolist(), g["grad_d_norms"].tolist()) if n > 1e-4), reverse=True)
    print(f"   loss_d {loss_d.item():.5f} vs {float(g['loss_d']):.5f}; D grad-norm dev worst 8:", [(f"{a:.2e}", f"{n:.2e}", k) for a, n, k in dev[:8]])

# synthesise
g = G("synth_small")
for mode in ("f32", "bf16"):
    precision.set_precision(mode)
    gen = make_generator(_small_cfg()).to("cuda").eval()
    W = S.make_weights(S.generator_schema(S.SMALL), int(g["seed"]))
    W["generator.duration_predictor.linear.bias"].fill_(float(g["dur_bias"]))
    gen.load_state_dict({k[len("generator."):]: v for k, v in W.items()})
    out = gen.synthesise(torch.from_numpy(g["in_x"]).cuda(), torch.from_numpy(g["in_x_lengths"]), d_factor=1.1, p_factor=1.6, e_factor=1.2)
    same = np.array_equal(out["durations"].numpy(), g["durations"])
    print(f"== synth_small [{mode}] durations equal {same}; wav rel {rel(out['wav'], g['wav']) if same else 'n/a'}; pitch rel {rel(out['pitch'], g['pitch']):.3e}")
precision.set_precision("f32")
