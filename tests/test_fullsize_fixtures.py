"""The CPU oracle pinned at the BENCHMARK'S OWN SIZE (B = 32, T_text <= 128, T_mel <= 800; 64-sentence synthesise; B = 32 Transformer)
against checksums the reference produced here (tools/make_golden_b32.py).  CPU-only; ~1 minute on 8 threads.

Integer paths exact; float checksums 1e-4 (both sides torch CPU f32: summation order only), gradient norms as in
tests/test_oracle_vs_golden.py (the kinked GAN losses amplify 1e-7 differences on the vocoder parameters)."""
import numpy as np
import torch

from oracle import generator as G
from oracle import schema as S
from tests import _golden_inputs as GI


def _t(batch):
    return {k: torch.from_numpy(v) for k, v in batch.items()}


def test_regenerated_inputs_match_the_generating_run(golden):
    g = golden("full_b32_gan")
    b = GI.gan_batch(g)
    assert b["mel"].shape == (32, 100, 800) and b["wav"].shape == (32, 800 * 256)
    sd, lens, x, Gc = GI.transformer_case(golden("full_b32_transformer"))
    assert x.shape == (32, 800, 256) and len(sd) == len(golden("full_b32_transformer")["keys"])
    # configs[3] as a whole model (round 5): same batch recipe, the reference module's own state-dict names / shapes stored
    gt = golden("full_b32_transformer_gan")
    bt = GI.gan_batch(gt)
    assert bt["mel"].shape == (32, 100, 800)
    names = gt["state_names"].tolist()
    assert any(k.startswith("encoder.transformer.encoders.3.self_attn.linear_q") for k in names) and len(names) == len(gt["state_shapes"])
    assert np.isfinite(float(gt["loss_g"])) and len(gt["grad_g_norms"]) > 100 and int((gt["durations"].sum(1) == gt["in_mel_lengths"]).sum()) == 32


def test_oracle_gan_step_at_the_benchmark_size(golden):
    g = golden("full_b32_gan")
    P = S.make_weights(S.generator_schema(S.Cfg()), int(g["seed"]))
    P.update(S.make_weights(S.discriminator_schema(), int(g["disc_seed"])))
    for v in P.values():
        v.requires_grad_(True)
    res = G.training_step(P, _t(GI.gan_batch(g)), rand01=torch.from_numpy(g["rand01"]), fb=None, with_mel=False, keep=True)
    out = res["out"]
    assert np.array_equal(out["start_idx"].numpy(), g["start_idx"])
    assert np.array_equal(out["durations"].numpy(), g["durations"])                    # 32 MAS paths + bincount: exact
    for k in ("loss", "align_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(float(out[k]) - float(g[k])) <= 1e-5 * abs(float(g[k])), k
    assert abs(res["wav"].double().norm().item() - g["wav_cks"][1]) <= 1e-9 * g["wav_cks"][1]
    w = out["wav_hat"].detach().double()
    assert abs(w.norm().item() - float(g["wav_hat_l2"])) <= 1e-4 * float(g["wav_hat_l2"])
    for k in ("loss_gen_mp", "loss_gen_mrd", "loss_fm_mp", "loss_fm_mrd", "mr_stft_loss"):
        assert abs(float(res["gen_logs"][k]) - float(g["genlog_" + k])) <= 1e-4 * abs(float(g["genlog_" + k])), k
    assert abs(float(res["loss_g"]) - float(g["loss_g"])) <= 1e-5 * abs(float(g["loss_g"]))
    assert abs(float(res["loss_d"]) - float(g["loss_d"])) <= 1e-5 * abs(float(g["loss_d"]))
    got = {k[len("generator."):]: v for k, v in res["grads_g"].items()}
    assert sorted(k for k, v in got.items() if v is None) == sorted(g["grad_g_none"].tolist())
    for k, n in zip(g["grad_g_names"].tolist(), g["grad_g_norms"].tolist()):
        tol = 1e-2 if k.startswith("vocoder.") else 5e-4
        assert abs(got[k].double().norm().item() - n) <= tol * max(n, 1e-6) + 1e-9, (k, got[k].double().norm().item(), n)
    gd = {k[len("discriminator."):]: v for k, v in res["grads_d"].items()}
    for k, n in zip(g["grad_d_names"].tolist(), g["grad_d_norms"].tolist()):
        assert abs(gd[k].double().norm().item() - n) <= 2e-3 * max(n, 1e-6) + 1e-9, k


def test_oracle_synthesise_64_sentences(golden):
    g = golden("full_b64_synth")
    P = S.make_weights(S.generator_schema(S.Cfg()), int(g["seed"]))
    P["generator.duration_predictor.linear.bias"].fill_(float(g["dur_bias"]))
    d, p, e = (float(v) for v in g["factors"])
    out = G.synthesise(P, torch.from_numpy(g["in_x"]), torch.from_numpy(g["in_x_lengths"]), d, p, e)
    assert np.array_equal(out["durations"].numpy(), g["durations"])                    # int64 exact, 64 x 128
    assert np.array_equal(out["wav_lengths"].numpy(), g["wav_lengths"])
    wav = out["wav"].double().numpy()
    assert tuple(wav.shape) == tuple(g["wav_shape"])
    step = int(g["wav_probe_step"])
    probe = wav[:, ::step][:, :257]
    scale = np.abs(g["wav_probe"]).max()
    assert np.abs(probe - g["wav_probe"]).max() <= 1e-3 * scale
    wl = g["wav_lengths"]
    l2 = np.array([np.sqrt((wav[b, :wl[b]] ** 2).sum()) for b in range(len(wl))])
    assert np.all(np.abs(l2 - g["wav_l2"]) <= 1e-4 * g["wav_l2"])


def test_oracle_transformer_at_the_benchmark_size(golden):
    from oracle import transformer as OT
    g = golden("full_b32_transformer")
    sd, lens, x, Gc = GI.transformer_case(g)
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd.items()}
    xt = torch.from_numpy(x).requires_grad_(True)
    T = x.shape[1]
    pad = torch.arange(T)[None] >= torch.from_numpy(lens)[:, None]
    y = OT.forward(P, xt, pad, heads=2)
    (y * torch.from_numpy(Gc)).sum().backward()
    valid = (~pad)[:, :, None]
    yv = (y.detach() * valid).double()
    assert abs(yv.norm().item() - g["y_cks"][1]) <= 1e-4 * g["y_cks"][1]
    assert np.abs((y.detach() * valid)[:, ::97, ::31].numpy() - g["y_probe"]).max() <= 1e-4 * np.abs(g["y_probe"]).max()
    dxv = (xt.grad * valid).double()
    assert abs(dxv.norm().item() - g["dx_cks"][1]) <= 1e-4 * g["dx_cks"][1]
    names = g["gnames"].tolist()
    for k, n in zip(names, g["gnorms"].tolist()):
        got = P[k].grad.double().norm().item()
        if n < 1e-3:                 # mathematically zero gradients (the key bias: softmax is shift-invariant) hold rounding noise only
            assert got < 1e-3, (k, got, n)
        else:
            assert abs(got - n) <= 1e-3 * n, (k, got, n)
