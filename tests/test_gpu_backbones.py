"""Other backbones: the depthwise-conv / dropout kernels against torch, and the LightSpeech
separable-conv encoder / decoder against fixtures produced by running the reference modules (tools/make_golden_backbones.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


@pytest.mark.parametrize("K_", [3, 5, 7, 9, 11, 13, 17, 21, 25, 31])
@pytest.mark.parametrize("B,T,C", [(3, 41, 64), (2, 130, 256), (1, 7, 384)])
def test_depthwise_conv_matches_conv1d_groups(K_, B, T, C):
    from optispeech_amd import kernels as K
    x, w, bias = rnd(B, T, C, seed=1), rnd(K_, C, seed=2, scale=0.3), rnd(C, seed=3)
    rm = (torch.rand(B * T, generator=torch.Generator().manual_seed(4)) > 0.25).float().to(DEV)
    xr = x.detach().cpu().double().transpose(1, 2).requires_grad_(True)                     # (B, C, T)
    wr = w.detach().cpu().double().t()[:, None, :].contiguous().requires_grad_(True)        # (C, 1, K)
    br = bias.detach().cpu().double().requires_grad_(True)
    yr = F.conv1d(xr, wr, br, padding=K_ // 2, groups=C) * rm.cpu().double().view(B, 1, T)
    y = K.dwconv_fwd(x, w, bias, rm)
    torch.testing.assert_close(y.cpu().double(), yr.detach().transpose(1, 2), rtol=1e-5, atol=1e-5)
    dy = rnd(B, T, C, seed=5)
    yr.backward(dy.cpu().double().transpose(1, 2))
    dx = K.dwconv_fwd(dy, w, None, rm, flip=True)
    torch.testing.assert_close(dx.cpu().double(), xr.grad.transpose(1, 2), rtol=1e-5, atol=1e-5)
    dw, db = rnd(K_, C, seed=6), rnd(C, seed=7)                                           # accumulate onto what is there
    dw0, db0 = dw.clone(), db.clone()
    K.dwconv_wgrad(dy, x, dw, db, rm)
    torch.testing.assert_close((dw - dw0).cpu().double(), wr.grad[:, 0, :].t(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close((db - db0).cpu().double(), br.grad, rtol=1e-4, atol=1e-4)


def test_dropout_add_forward_backward_share_the_mask():
    from optispeech_amd import ops
    x = (rnd(8, 50, 64, seed=1).abs() + 0.5).requires_grad_(True)
    res = rnd(8, 50, 64, seed=2).requires_grad_(True)
    p = 0.3
    y = ops.DropoutAddFn.apply(x, res, p, 4321, 9)
    ratio = ((y - res) / x).detach()
    keep = ratio > 0.5
    assert torch.all(keep | (ratio.abs() < 1e-6)) and torch.allclose(ratio[keep], torch.full_like(ratio[keep], 1 / (1 - p)), rtol=1e-5)
    assert abs((~keep).float().mean().item() - p) < 0.02
    g = rnd(8, 50, 64, seed=3)
    y.backward(g)
    torch.testing.assert_close(x.grad, g * ratio, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res.grad, g)
    y2 = ops.DropoutAddFn.apply(x, res, p, 4322, 9)                                        # another step: another mask
    assert not torch.equal(y2, y)


def _load(module, g):
    sd = {str(k): torch.from_numpy(g["w_" + str(k)]) for k in g["keys"]}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert sorted(module.state_dict().keys()) == sorted(sd.keys())                          # reference schema out as well
    for k, v in module.state_dict().items():
        assert v.shape == sd[k].shape, k


def _check(module, g, tol_y=2e-4, tol_g=2e-3):
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    lens = torch.from_numpy(g["lens"]).to(DEV)
    pad = torch.arange(x.shape[1], device=DEV)[None] >= lens[:, None]
    y = module(x, pad)
    torch.testing.assert_close(y.detach().cpu(), torch.from_numpy(g["y"]), rtol=tol_y, atol=tol_y)
    (y * torch.from_numpy(g["G"]).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-20)).item()      # noqa: E731
    assert rel(x.grad.cpu(), torch.from_numpy(g["dx"])) < tol_g
    # gradients: compare through the modules' own reference-layout mapping
    from optispeech_amd.model.base import RefSchemaModule
    for name, sub in module.named_modules():
        for pname, p in sub._parameters.items():
            if p is None:
                continue
            key, _, to_ref = sub._ref(pname) if isinstance(sub, RefSchemaModule) else (pname, None, None)
            full = (name + "." if name else "") + key
            want = torch.from_numpy(g["g_" + full])
            got = p.grad.detach().cpu()
            got = to_ref(got) if to_ref else got
            assert got.shape == want.shape, full
            # absolute floor: some gradients are analytically zero (a key bias shifts every score of a row alike) and are pure
            # round-off on both sides
            err = (got.double() - want.double()).norm().item()
            assert err < tol_g * want.double().norm().item() + 2e-5, (full, rel(got, want), err)


@pytest.mark.parametrize("which", ["enc", "dec"])
def test_lightspeech_backbone_vs_reference_golden(golden, which):
    from optispeech_amd import precision
    from optispeech_amd.model.lightspeech import LightSpeechTransformerDecoder, LightSpeechTransformerEncoder
    precision.set_precision("f32")
    g = golden("lightspeech_" + which)
    if which == "enc":
        m = LightSpeechTransformerEncoder(dim=64, kernel_sizes=[5, 25, 13, 9], activation="relu", dropout=0.2)
    else:
        m = LightSpeechTransformerDecoder(dim=64, kernel_sizes=[17, 21, 9, 13], activation="relu", dropout=0.2, max_source_positions=2000)
    m = m.to(DEV).eval()
    _load(m, g)
    _check(m, g)


def test_lightspeech_training_mode_runs_and_dropout_changes_per_step():
    from optispeech_amd import precision, rng
    from optispeech_amd.model.lightspeech import LightSpeechTransformerEncoder
    precision.set_precision("bf16")
    try:
        rng.reset_streams(); rng.manual_seed(7, 0)
        m = LightSpeechTransformerEncoder(dim=256, kernel_sizes=[5, 25, 13, 9], dropout=0.2).to(DEV).train()
        x = rnd(4, 128, 256, seed=1).requires_grad_(True)
        pad = torch.arange(128, device=DEV)[None] >= torch.tensor([128, 90, 64, 10], device=DEV)[:, None]
        y1 = m(x, pad)
        y1.square().mean().backward()
        assert torch.isfinite(y1).all() and torch.isfinite(x.grad).all()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        y1b = m(x, pad)
        assert torch.equal(y1, y1b)                                  # same step seed: same masks
        rng.advance()
        assert not torch.equal(m(x, pad), y1)
        assert torch.all(y1[pad] == 0)                               # the encoder's final mask
    finally:
        precision.set_precision("f32")


def test_lightspeech_generator_train_step_and_synthesise_run():
    """the LightSpeech pair wired into the full model (ModelConfig(backbone="lightspeech")): one GAN training step with finite
    losses, gradients reach the depthwise taps of both backbones, the optimizer moves them, synthesise() works"""
    from optispeech_amd import precision, rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    from optispeech_amd.values import InferenceInputs
    precision.set_precision("bf16")
    try:
        torch.manual_seed(2)
        rng.manual_seed(2, 0)
        cfg = ModelConfig(backbone="lightspeech")
        m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to(DEV).train()
        batch = synthetic_batch(2, 24, 96, cfg, seed=5, device=DEV)
        m.optimizers()
        for sch in m.lr_schedulers():                                 # no warm-up: the first step already moves the weights
            sch.warmup = 0
            sch.opt.lr = sch.base_lr
        w_enc = m.generator.encoder.layers[1].dw1
        w_dec = m.generator.decoder.layers[0].dw2
        before = (w_enc.detach().clone(), w_dec.detach().clone())
        m.training_step(batch, 0)
        logs = m.fetch_logs()
        assert all(np.isfinite(v) for v in logs.values()), logs
        torch.cuda.synchronize()
        assert w_enc.shape == (25, cfg.dim) and w_dec.shape == (17, cfg.dim)
        assert not torch.equal(w_enc.detach(), before[0]) and not torch.equal(w_dec.detach(), before[1])
        x = torch.randint(1, 150, (2, 16))
        xl = torch.tensor([16, 9])
        out = m.eval().synthesise(InferenceInputs(clean_text="", x=x * (torch.arange(16)[None] < xl[:, None]), x_lengths=xl,
                                                  d_factor=1.0, p_factor=1.0, e_factor=1.0),
                                  durations_override=torch.full((2, 16), 3))
        wav = torch.as_tensor(out.wav)
        assert wav.shape[0] == 2 and torch.isfinite(wav).all()
    finally:
        precision.set_precision("f32")


@pytest.mark.parametrize("B,T,H", [(3, 5, 64), (8, 50, 128), (5, 1, 64), (40, 23, 256), (32, 64, 256)])
def test_lstm_matches_torch_lstm(B, T, H):
    """csrc/lstm.hip (team-of-workgroups recurrence) + the GEMMs around it against torch.nn.LSTM on the CPU: outputs, input
    gradient and all four parameter gradients; batch sizes that need zero padding (B % 8 != 0) and chunking (B * H/32 > 256)."""
    from optispeech_amd import ops, precision
    precision.set_precision("f32")
    torch.manual_seed(B * 1000 + T)
    ref = torch.nn.LSTM(H, H, num_layers=1, batch_first=True).double()
    x = torch.randn(B, T, H, dtype=torch.float64, requires_grad=True)
    g = torch.randn(B, T, H, dtype=torch.float64)
    yr, _ = ref(x)
    (yr * g).sum().backward()
    ps = [torch.nn.Parameter(getattr(ref, n).detach().float().to(DEV)) for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    xd = x.detach().float().to(DEV).requires_grad_(True)
    y = ops.lstm(xd, *ps)
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=2e-4, atol=2e-5)
    (y * g.float().to(DEV)).sum().backward()
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-20)).item()      # noqa: E731
    assert rel(xd.grad, x.grad) < 1e-3
    for p, n in zip(ps, ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")):
        assert rel(p.grad, getattr(ref, n).grad) < 1e-3, n


def test_leanspeech_backbone_vs_reference_golden(golden):
    from optispeech_amd import precision
    from optispeech_amd.model.leanspeech import LeanSpeechBackbone
    precision.set_precision("f32")
    g = golden("leanspeech")
    m = LeanSpeechBackbone(dim=64, kernel_size=9, num_layers=2, drop_path=0.2).to(DEV).eval()
    _load(m, g)
    _check(m, g)


def test_leanspeech_generator_train_step_runs():
    """ModelConfig(backbone="leanspeech"): one GAN training step at the BASELINE width (H = 256: 8 workgroups per utterance),
    finite losses, the recurrent weights of both backbones move."""
    from optispeech_amd import precision, rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    precision.set_precision("bf16")
    try:
        torch.manual_seed(2)
        rng.manual_seed(2, 0)
        cfg = ModelConfig(backbone="leanspeech")
        m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to(DEV).train()
        batch = synthetic_batch(2, 24, 96, cfg, seed=5, device=DEV)
        m.optimizers()
        for sch in m.lr_schedulers():
            sch.warmup = 0
            sch.opt.lr = sch.base_lr
        w_e, w_d = m.generator.encoder.layers[0].w_hh, m.generator.decoder.layers[3].w_hh
        before = (w_e.detach().clone(), w_d.detach().clone())
        m.training_step(batch, 0)
        logs = m.fetch_logs()
        assert all(np.isfinite(v) for v in logs.values()), logs
        torch.cuda.synchronize()
        assert not torch.equal(w_e.detach(), before[0]) and not torch.equal(w_d.detach(), before[1])
    finally:
        precision.set_precision("f32")


def test_conformer_backbone_vs_reference_golden(golden):
    """relative-position attention (q + u / q + v products, rel-shift), macaron feed-forward pair, convolution module with
    BatchNorm running statistics, swish / GLU: output, input gradient and every parameter gradient vs the reference run"""
    from optispeech_amd import precision
    from optispeech_amd.model.conformer import Conformer
    precision.set_precision("f32")
    g = golden("conformer")
    m = Conformer(dim=64, attention_heads=2, linear_units=96, num_blocks=2, cnn_module_kernel=7).to(DEV).eval()
    _load(m, g)
    _check(m, g)


def test_conformer_generator_train_step_runs():
    """ModelConfig(backbone="conformer") (cnn kernel 7 / 31 as configs/model/generator/{encoder,decoder}/conformer.yaml): one GAN
    training step, finite losses, BatchNorm statistics updated, relative-position biases move."""
    from optispeech_amd import precision, rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    precision.set_precision("bf16")
    try:
        torch.manual_seed(2)
        rng.manual_seed(2, 0)
        cfg = ModelConfig(backbone="conformer")
        m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to(DEV).train()
        batch = synthetic_batch(2, 24, 96, cfg, seed=5, device=DEV)
        m.optimizers()
        for sch in m.lr_schedulers():
            sch.warmup = 0
            sch.opt.lr = sch.base_lr
        att = m.generator.decoder.conformer.encoders[0].self_attn
        bn = m.generator.decoder.conformer.encoders[0].conv_module.norm
        assert m.generator.decoder.conformer.encoders[0].conv_module.depthwise_conv.weight.shape[0] == 31
        before = att.pos_bias_u.detach().clone()
        m.training_step(batch, 0)
        logs = m.fetch_logs()
        assert all(np.isfinite(v) for v in logs.values()), logs
        torch.cuda.synchronize()
        assert not torch.equal(att.pos_bias_u.detach(), before)
        assert int(bn.num_batches_tracked) == 1 and bn.running_mean.abs().sum().item() > 0
    finally:
        precision.set_precision("f32")
