"""BASELINE-size (B = 32 per GPU, 64-frame segments = 16 384 samples, both waves = 64 sequences) checks of the discriminator
kernels through size-independent properties: the three convolution kernels of a layer are mutually adjoint
(<dy, conv(x; w)> = <dgrad(dy; w), x> = <wgrad(dy, x), w>), sequences are independent (a sub-batch reproduces its rows
bit-exactly: no leakage across utterance / period-column boundaries, all tiles and the XCD tile order covered), and the
linear first layer is linear.  Tolerances are those of bf16 operands with f32 accumulation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dot(a, b):
    return (a.double().flatten() * b.double().flatten()).sum().item()


@pytest.mark.parametrize("name,U,H,W,cin,cout,spec", [
    ("MPD conv4 p=5", 64 * 5, 41, 1, 1024, 1024, (5, 1, 1, 1, 2, 0)),        # DiscriminatorP convs[4], period 5, 2B waves
    ("MPD conv3 p=3", 64 * 3, 203, 1, 512, 1024, (5, 1, 3, 1, 2, 0)),        # strided (3,1)
    ("MRD conv1 1024", 64, 33, 257, 64, 64, (3, 5, 1, 2, 1, 2)),             # DiscriminatorR convs[1] (k(5,3) s(2,1) upstream)
    ("MRD conv2 512", 64, 65, 65, 64, 64, (3, 5, 2, 2, 1, 2)),
    ("MPD post p=11", 64 * 11, 19, 1, 1024, 1, (3, 1, 1, 1, 1, 0)),          # conv_post: row-dot / outer-product kernels
])
def test_layer_kernels_are_mutually_adjoint(name, U, H, W, cin, cout, spec):
    from optispeech_amd import disc_ops as D, kernels as K
    KH, KW, sh, sw, ph, pw = spec
    g = torch.Generator().manual_seed(0)
    x = torch.randn(U, H, W, cin, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(cout, KH, KW, cin, generator=g) / np.sqrt(cin * KH * KW)).to(DEV)
    wb = K.cast_bf16(w)
    y = D.conv2d_fwd(x, wb, None, KH, KW, sh, sw, ph, pw, None, False)                 # f32 out, no bias / activation
    dy = torch.randn(y.shape, generator=g).to(DEV).to(torch.bfloat16)
    dx = D.conv2d_dgrad(dy, D.transpose_weight2d(wb), H, W, KH, KW, sh, sw, ph, pw)
    dw, db = D.conv2d_wgrad(dy, x, KH, KW, sh, sw, ph, pw)
    a, b, c = _dot(dy, y), _dot(dx, x), _dot(dw, wb.float())
    scale = np.sqrt(_dot(dy, dy) * _dot(y, y))
    assert abs(a - b) <= 2e-3 * scale and abs(a - c) <= 2e-3 * scale, (name, a, b, c, scale)
    assert torch.allclose(db, dy.float().sum((0, 1, 2)), rtol=1e-3, atol=1e-2 * dy.float().abs().sum().item() / dy.numel() * np.sqrt(dy.numel() / cout))


@pytest.mark.parametrize("which,sub", [("multiperioddisc", 3), ("multiresddisc", 17)])
def test_full_batch_rows_equal_single_sequence(which, sub):
    """forward of the whole 64-wave batch vs the same wave alone: identical scores and feature maps"""
    from optispeech_amd import precision
    from optispeech_amd.config import ModelConfig, make_optispeech
    precision.set_precision("bf16")
    try:
        torch.manual_seed(0)
        m = make_optispeech(ModelConfig()).to(DEV).eval()
        disc = getattr(m.discriminator, which)
        wav = torch.randn(64, 16384, generator=torch.Generator().manual_seed(1)).clamp(-1, 1).to(DEV)
        with torch.no_grad():
            for d in disc.discriminators:
                o, fm = d(wav)
                o1, fm1 = d(wav[sub:sub + 1])
                assert torch.equal(o[sub:sub + 1], o1)
                per = fm[0].shape[0] // 64
                for f, f1 in zip(fm, fm1):
                    assert torch.equal(f[sub * per:(sub + 1) * per], f1)
    finally:
        precision.set_precision("f32")


def test_first_layer_is_linear_at_full_size():
    from optispeech_amd import kernels as K
    U, H, W, cout, KH, KW, sh, sw, ph, pw = 64, 65, 513, 64, 7, 5, 2, 2, 3, 2
    g = torch.Generator().manual_seed(2)
    x1, x2 = torch.randn(U, H, W, generator=g).to(DEV), torch.randn(U, H, W, generator=g).to(DEV)
    w = (torch.randn(cout, KH * KW, generator=g) * 0.1).to(DEV)
    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    f = lambda x: K.smallcin_fwd(x, w, None, U=U, Hin=H, Win=W, Ho=Ho, Wo=Wo, cout=cout, KH=KH, KW=KW, sh=sh, sw=sw, ph=ph,   # noqa: E731
                                 pw=pw, slope=None, out_bf16=True).float()
    lhs, rhs = f(0.5 * x1 - 2.0 * x2), 0.5 * f(x1) - 2.0 * f(x2)
    assert (lhs - rhs).abs().max().item() <= 2e-2 * rhs.abs().max().item()


def test_full_size_training_step_is_finite_and_consistent():
    """one BASELINE-shaped bf16 step: every logged scalar finite, both gradient arenas finite and non-zero"""
    from optispeech_amd import precision, rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    precision.set_precision("bf16")
    try:
        torch.manual_seed(1234)
        rng.manual_seed(1234, 0)
        cfg = ModelConfig()
        m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to(DEV).train()
        batch = synthetic_batch(32, 128, 800, cfg, seed=1234, device=DEV)
        og, od = m.optimizers()
        m.training_step(batch, 0)
        logs = m.fetch_logs()
        assert all(np.isfinite(v) for v in logs.values()), logs
        for o in (og, od):
            assert torch.isfinite(o.arena.grad).all() and o.arena.grad.abs().sum().item() > 0
            assert torch.isfinite(o.arena.data).all()
    finally:
        precision.set_precision("f32")
