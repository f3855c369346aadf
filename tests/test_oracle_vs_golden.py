"""Pin the CPU oracle against fixtures produced by running the reference (tools/make_golden.py).

CPU-only.  Tolerances: index / integer paths exact; floating point 1e-5 relative to the tensor's
scale unless noted (both sides are torch CPU fp32, differences are summation-order only).
"""
import numpy as np
import pytest
import torch

from oracle import alignment as A
from oracle import disc as D
from oracle import generator as G
from oracle import losses as Ls
from oracle import nn_ops as N
from oracle import schema as S


def close(a, b, rtol=1e-5, atol_scale=1e-5):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    assert torch.equal(fa, fb), "non-finite pattern differs"
    assert torch.equal(a[~fa], b[~fb]) or (~fa).sum() == 0
    a, b = a[fa], b[fb]
    if a.numel() == 0:
        return
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item()
    assert err <= atol_scale * scale + rtol * scale, f"max err {err:g} vs scale {scale:g}"


def batch_of(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}


# ----------------------------------------------------------------------------- units
def test_mas_paths_exact(golden):
    g = golden("units")
    i = 0
    while f"mas{i}_lp" in g.files:
        lp, want = g[f"mas{i}_lp"], g[f"mas{i}_path"]
        assert np.array_equal(A.mas_path_np(lp), want)
        assert np.array_equal(A.mas_path_c(lp), want)
        i += 1
    assert i >= 5


def test_prior_tables(golden):
    g = golden("units")
    tl, fl = torch.from_numpy(g["prior_tl"]), torch.from_numpy(g["prior_fl"])
    close(A.batched_prior(tl, fl), g["prior"], 1e-6, 1e-6)
    # the lgamma-table closed form (what the device evaluates) agrees with scipy to fp32 rounding
    for T, N in ((11, 5), (20, 7), (800, 128)):
        a = A.betabinom_prior_np(T, N).astype(np.float32)
        b = A.betabinom_prior_lgamma_np(T, N).astype(np.float32)
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(a))) < 2e-6


def test_average_expand_upsample(golden):
    g = golden("units")
    out = A.average_by_duration(torch.from_numpy(g["abd_ds"]), torch.from_numpy(g["abd_xs"][..., 0]),
                                torch.tensor([5, 2]), torch.tensor([9, 6]))
    close(out, g["abd_out"], 1e-6, 1e-6)
    ex, ln = A.expand_by_duration(torch.from_numpy(g["exp_x"]), torch.from_numpy(g["exp_dur"]))
    assert np.array_equal(ex.numpy(), g["exp_out"]) and np.array_equal(ln.numpy(), g["exp_len"])
    gu = A.gaussian_upsampling(torch.from_numpy(g["gu_hs"]), torch.from_numpy(g["exp_dur"]).float(),
                               torch.from_numpy(g["gu_hm"]), torch.from_numpy(g["gu_dm"]))
    close(gu, g["gu_out"])


def test_regression_and_forwardsum_losses(golden):
    g = golden("units")
    d, p, e = (torch.from_numpy(g[k][..., 0]) for k in ("fs2_d", "fs2_p", "fs2_e"))
    ds = torch.from_numpy(g["abd_ds"])
    ps, es = torch.from_numpy(g["fs2_ps"][..., 0]), torch.from_numpy(g["fs2_es"][..., 0])
    out = Ls.variance_losses(d, p, e, ds, ps, es, torch.from_numpy(g["fs2_il"]))
    close(torch.stack(out), g["fs2_out"])
    lp = torch.from_numpy(g["fsl_lp"]).requires_grad_(True)
    loss = Ls.forward_sum_loss(lp, torch.tensor([5, 3]), torch.tensor([12, 9]))
    loss.backward()
    close(loss.detach(), g["fsl_out"])
    close(lp.grad, g["fsl_grad"])


def test_mr_stft_loss(golden):
    g = golden("units")
    x = torch.from_numpy(g["stft_x"]).requires_grad_(True)
    sc, mag = Ls.mr_stft_loss(x, torch.from_numpy(g["stft_y"]))
    (sc + mag).backward()
    close(sc.detach(), g["stft_sc"])
    close(mag.detach(), g["stft_mag"])
    close(x.grad, g["stft_grad"], 1e-4, 1e-4)


# ----------------------------------------------------------------------------- generator
def _leaf_params(schema, seed, grad_prefix=None):
    P = S.make_weights(schema, seed)
    for k, v in P.items():
        if grad_prefix and k.startswith(grad_prefix):
            v.requires_grad_(True)
    return P


@pytest.mark.parametrize("name,cfg,full", [("gen_small_am", S.SMALL, True), ("gen_full_b2", S.Cfg(), False)])
def test_generator_forward_backward(golden, name, cfg, full):
    g = golden(name)
    P = _leaf_params(S.generator_schema(cfg), int(g["seed"]), "generator.")
    batch = batch_of(g)
    out = G.generator_forward(P, batch, rand01=torch.from_numpy(g["rand01"]), keep=True)
    assert np.array_equal(out["start_idx"].numpy(), g["start_idx"])
    assert np.array_equal(out["durations"].numpy(), g["durations"])          # MAS + bincount: exact
    for k in ("p_avg", "e_avg", "d_hat", "p_hat", "e_hat"):
        close(out[k].detach(), g[k], 1e-4, 1e-4)
    for k in ("loss", "align_loss", "duration_loss", "pitch_loss", "energy_loss", "bin_loss", "forwardsum_loss"):
        close(out[k].detach(), g[k], 1e-5, 1e-5)
    for k in ("text_emb", "enc", "log_p_attn", "xp", "xe", "y_up", "dec", "wav_hat"):
        v = out[k].detach()
        if full:
            close(v, g[k], 1e-4, 1e-4)
        else:
            fin = torch.where(torch.isfinite(v), v, torch.zeros_like(v)).double()
            assert abs(fin.norm().item() - float(g[k + "_l2"])) <= 1e-4 * float(g[k + "_l2"])
    grads = torch.autograd.grad(out["loss"], [P[k] for k in P], allow_unused=True)
    got = {k[len("generator."):]: gr for k, gr in zip(P, grads)}
    none = sorted(k for k, v in got.items() if v is None)
    assert none == sorted(g["grad_g_none"].tolist())                          # decoder.*, vocoder.*, energy embed
    for k, n in zip(g["grad_g_names"].tolist(), g["grad_g_norms"].tolist()):
        assert abs(got[k].double().norm().item() - n) <= 2e-4 * max(n, 1e-6) + 1e-9, k
    if full:
        for key in g.files:
            if key.startswith("grad_g/"):
                close(got[key[len("grad_g/"):]], g[key], 2e-4, 2e-4)


def test_gan_training_step(golden):
    g = golden("gen_small_gan")
    P = _leaf_params(S.generator_schema(S.SMALL), int(g["seed"]), "generator.")
    P.update(_leaf_params(S.discriminator_schema(), 4321, "discriminator."))
    batch = batch_of(g)
    res = G.training_step(P, batch, rand01=torch.from_numpy(g["rand01"]), fb=None, with_mel=False)
    close(res["wav"], g["wav"], 0, 0)
    close(res["out"]["wav_hat"].detach(), g["wav_hat"], 1e-4, 1e-4)
    close(res["gen_adv_loss"].detach(), g["gen_adv_loss"], 1e-5, 1e-5)
    for k in ("loss_gen_mp", "loss_gen_mrd", "loss_fm_mp", "loss_fm_mrd", "mr_stft_loss", "sc", "mag"):
        close(res["gen_logs"][k].detach(), g["genlog_" + k], 1e-5, 1e-5)
    close(res["loss_g"], g["loss_g"], 1e-5, 1e-5)
    close(res["loss_d"], g["loss_d"], 1e-5, 1e-5)
    got = {k[len("generator."):]: v for k, v in res["grads_g"].items()}
    assert sorted(k for k, v in got.items() if v is None) == sorted(g["grad_g_none"].tolist())
    # GAN-phase gradients pass through |STFT|->log, hinge, LeakyReLU and clip: summation-order noise of 1e-7 in
    # wav_hat moves a few kinks, so vocoder grads agree to ~1e-3 only (acoustic-model grads stay at 1e-4).
    for k, n in zip(g["grad_g_names"].tolist(), g["grad_g_norms"].tolist()):
        tol = 1e-2 if k.startswith("vocoder.") else 5e-4
        assert abs(got[k].double().norm().item() - n) <= tol * max(n, 1e-6) + 1e-9, k
    gd = {k[len("discriminator."):]: v for k, v in res["grads_d"].items()}
    for k, n in zip(g["grad_d_names"].tolist(), g["grad_d_norms"].tolist()):
        assert abs(gd[k].double().norm().item() - n) <= 2e-3 * max(n, 1e-6) + 1e-9, k
    for key in g.files:
        if key.startswith("grad_d/"):
            close(gd[key[len("grad_d/"):]], g[key], 2e-3, 2e-3)


def test_synthesise(golden):
    g = golden("synth_small")
    P = S.make_weights(S.generator_schema(S.SMALL), int(g["seed"]))
    P["generator.duration_predictor.linear.bias"].fill_(float(g["dur_bias"]))
    out = G.synthesise(P, torch.from_numpy(g["in_x"]), torch.from_numpy(g["in_x_lengths"]), 1.1, 1.6, 1.2)
    assert np.array_equal(out["durations"].numpy(), g["durations"])           # int64 exact
    assert np.array_equal(out["wav_lengths"].numpy(), g["wav_lengths"])
    close(out["pitch"], g["pitch"], 1e-4, 1e-4)
    close(out["energy"], g["energy"], 1e-4, 1e-4)
    close(out["wav"], g["wav"], 1e-3, 1e-3)


def test_mel_filterbank_shape_and_partition():
    """Parity unpinned (torchaudio absent): structural checks only."""
    fb = Ls.mel_filterbank(22050, 1024, 100, 80, 8000)
    assert fb.shape == (513, 100) and bool((fb >= 0).all())
    peak = fb.argmax(0)
    assert bool((peak[1:] >= peak[:-1]).all())


def test_transformer_oracle_vs_reference_golden(golden):
    """A19: the restated Transformer backbone vs the reference module's output and gradients (ragged batch, 2 blocks)."""
    from oracle import transformer as OT
    g = golden("transformer")
    P = {k: torch.from_numpy(g["w_" + k]).requires_grad_(True) for k in g["keys"].tolist()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    lens = torch.from_numpy(g["lens"])
    pad = torch.arange(x.shape[1])[None] >= lens[:, None]
    y = OT.forward(P, x, pad, heads=2)
    assert torch.allclose(y, torch.from_numpy(g["y"]), rtol=1e-5, atol=1e-5)
    (y * torch.from_numpy(g["G"])).sum().backward()
    assert torch.allclose(x.grad, torch.from_numpy(g["dx"]), rtol=1e-4, atol=1e-5)
    for k in g["keys"].tolist():
        assert torch.allclose(P[k].grad, torch.from_numpy(g["g_" + k]), rtol=1e-4, atol=2e-5), k
