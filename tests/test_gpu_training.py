"""STFT kernels, optimiser kernel and the GAN training step on the GPU vs the oracle / reference goldens."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import disc as OD                       # noqa: E402  (checker only)
from oracle import generator as OG                  # noqa: E402
from oracle import losses as OL                     # noqa: E402
from oracle import schema as S                      # noqa: E402

DEV = "cuda"


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# ------------------------------------------------------------------------------------------------ STFT
@pytest.mark.parametrize("n_fft,hop,win,rect,clamp", [(1024, 120, 600, False, 1e-7), (2048, 240, 1200, False, 1e-7),
                                                      (512, 50, 240, False, 1e-7), (1024, 256, 1024, True, None),
                                                      (2048, 512, 2048, True, None), (512, 128, 512, True, None),
                                                      (1024, 256, 1024, False, None)])
def test_stft_magnitude_forward_backward(n_fft, hop, win, rect, clamp):
    from optispeech_amd import spectral
    B, T = 3, 16384
    g = torch.Generator().manual_seed(n_fft + hop)
    x = (torch.rand(B, T, generator=g) * 2 - 1).requires_grad_(True)
    w = torch.ones(n_fft) if rect else torch.hann_window(win)
    want = OL.stft_mag(x, n_fft, hop, n_fft if rect else win, w, clamp)
    dm = torch.randn(want.shape, generator=g)
    (want * dm).sum().backward()
    xg = x.detach().to(DEV).requires_grad_(True)
    got = spectral.stft_magnitude(xg, n_fft, hop, None if rect else win, None if rect else w.to(DEV), clamp)
    assert got.shape == want.shape
    assert relerr(got, want) < 2e-5
    (got * dm.to(DEV)).sum().backward()
    assert relerr(xg.grad, x.grad) < 2e-4


def test_mr_stft_and_mel_losses_vs_oracle(golden):
    from optispeech_amd import spectral
    g = golden("units")
    x = torch.from_numpy(g["stft_x"]).to(DEV).requires_grad_(True)
    y = torch.from_numpy(g["stft_y"]).to(DEV)
    sc, mag = spectral.MultiResolutionSTFTLoss().to(DEV)(x, y)
    (sc + mag).backward()
    assert abs(sc.item() - float(g["stft_sc"])) < 1e-5 * float(g["stft_sc"])          # reference golden
    assert abs(mag.item() - float(g["stft_mag"])) < 1e-5 * float(g["stft_mag"])
    assert relerr(x.grad, g["stft_grad"]) < 1e-3
    # mel loss: oracle restatement (parity unpinned against torchaudio, see oracle/losses.py)
    fb = OL.mel_filterbank(22050, 1024, 100, 80, 8000)
    xc = torch.from_numpy(g["stft_x"]).requires_grad_(True)
    want = OL.mel_l1_loss(xc, torch.from_numpy(g["stft_y"]), fb)
    want.backward()
    m = spectral.MelSpecReconstructionLoss(22050, 1024, 256, 1024, 100, 80, 8000).to(DEV)
    x2 = torch.from_numpy(g["stft_x"]).to(DEV).requires_grad_(True)
    got = m(x2, y)
    got.backward()
    assert abs(got.item() - want.item()) < 1e-5 * want.item()
    assert relerr(x2.grad, xc.grad) < 1e-3


# ------------------------------------------------------------------------------------------------ optimiser
def test_fused_adamw_matches_torch():
    from optispeech_amd.optim import FusedAdamW
    torch.manual_seed(0)
    shapes = [(37, 5), (256,), (3, 3, 7), (1,), (1000, 13)]
    ref = [torch.randn(s).requires_grad_(True) for s in shapes]
    mine = [torch.nn.Parameter(r.detach().clone().to(DEV)) for r in ref]
    opt_r = torch.optim.AdamW(ref, lr=2e-4, betas=(0.8, 0.99), weight_decay=1e-2)
    opt_m = FusedAdamW(mine, lr=2e-4, betas=(0.8, 0.99), weight_decay=1e-2)
    for it in range(4):
        grads = [torch.randn(s) * (10.0 if it == 1 else 0.1) for s in shapes]
        for r, g in zip(ref, grads):
            r.grad = g.clone()
        opt_m.zero_grad()
        for m, g in zip(mine, grads):
            m.grad.add_(g.to(DEV))
        torch.nn.utils.clip_grad_norm_(ref, 10.0)
        opt_r.step()
        opt_m.step(max_norm=10.0)
        for r, m in zip(ref, mine):
            assert relerr(m, r) < 1e-5


# ------------------------------------------------------------------------------------------------ GAN step
def _small_model(g):
    from optispeech_amd.config import ModelConfig, make_optispeech
    c = S.SMALL
    cfg = ModelConfig(dim=c.dim, enc_inter=c.enc_inter, dec_inter=c.dec_inter, dur=c.dur + (0.0,), pitch=c.pitch + (0.0,),
                      energy=c.energy + (0.0,), voc_dim=c.voc_dim, voc_inter=c.voc_inter,
                      voc_layers=c.voc_layers).no_dropout()
    m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to(DEV).train()
    W = S.make_weights(S.generator_schema(S.SMALL), int(g["seed"]))
    W.update(S.make_weights(S.discriminator_schema(), 4321))
    missing, unexpected = m.load_state_dict(W, strict=False)
    assert not unexpected and all(("melspec" in k or "window" in k) for k in missing), (missing, unexpected)
    m.generator.segment_rand01 = torch.from_numpy(g["rand01"])
    return m


def _ref_grads(module, prefix=""):
    out = {}
    for pname, p in module.named_parameters():
        mn, lf = pname.rsplit(".", 1)
        mod = module.get_submodule(mn)
        key, _, to_ref = mod._ref(lf) if hasattr(mod, "_ref") else (lf, None, None)
        gr = p.grad
        out[prefix + mn + "." + key] = None if gr is None else (to_ref(gr) if to_ref else gr)
    return out


def test_gan_training_step_vs_reference_golden(golden):
    """G phase (AM + adversarial/FM/MR-STFT losses, mel term off as in the golden) and D phase against values the
    reference produced; then one full optimiser step is sanity-checked."""
    g = golden("gen_small_gan")
    m = _small_model(g)
    batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    batch.update(sids=None, lids=None)
    m.discriminator.lambda_mel = 0.0                       # the reference golden could not run torchaudio's mel
    logs = {}
    for p in m.discriminator.parameters():
        p.requires_grad_(False)
    loss_g, (wav, wav_hat) = m.training_step_g(batch, True, logs)
    assert relerr(wav, g["wav"]) == 0.0                                                   # index-exact segment gather
    assert relerr(wav_hat, g["wav_hat"]) < 1e-3
    for k in ("loss_gen_mp", "loss_gen_mrd", "loss_fm_mp", "loss_fm_mrd", "mr_stft_loss"):
        assert abs(logs["gen_adv_loss/train_" + k].item() - float(g["genlog_" + k])) <= 2e-4 * abs(float(g["genlog_" + k])), k
    assert abs(loss_g.item() - float(g["loss_g"])) <= 1e-4 * abs(float(g["loss_g"]))
    loss_g.backward()
    got = _ref_grads(m.generator)
    for k, n in zip(g["grad_g_names"].tolist(), g["grad_g_norms"].tolist()):
        tol = 2e-2 if k.startswith("vocoder.") else 2e-3          # kinked GAN losses: see tests/test_oracle_vs_golden.py
        assert abs(got[k].double().norm().item() - n) <= tol * max(n, 1e-6) + 1e-8, (k, got[k].double().norm().item(), n)
    for p in m.discriminator.parameters():
        p.requires_grad_(True)
        p.grad = None
    loss_d = m.training_step_d(batch, (wav, wav_hat.detach()), logs)
    assert abs(loss_d.item() - float(g["loss_d"])) <= 1e-4 * abs(float(g["loss_d"]))
    loss_d.backward()
    gd = _ref_grads(m.discriminator)
    for k, n in zip(g["grad_d_names"].tolist(), g["grad_d_norms"].tolist()):
        assert abs(gd[k].double().norm().item() - n) <= 5e-3 * max(n, 1e-6) + 1e-8, k


def test_training_step_runs_and_updates():
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    c = S.SMALL
    cfg = ModelConfig(dim=c.dim, enc_inter=c.enc_inter, dec_inter=c.dec_inter, dur=c.dur + (0.1,), pitch=c.pitch + (0.5,),
                      energy=c.energy + (0.5,), voc_dim=c.voc_dim, voc_inter=c.voc_inter, voc_layers=c.voc_layers)
    torch.manual_seed(0)
    m = make_optispeech(cfg, batch_size=4, pretraining_steps=2).to(DEV).train()
    batch = synthetic_batch(4, 32, 160, cfg, ragged=True, device=DEV)
    w0 = m.generator.encoder.convnext[0].pwconv1_weight.detach().clone()
    d0 = m.discriminator.multiperioddisc.discriminators[0].convs[1].weight_v.detach().clone()
    dec0 = m.generator.decoder.convnext[0].pwconv1_weight.detach().clone()
    for i in range(4):
        m.training_step(batch, i)
        logs = m.fetch_logs()
        assert all(np.isfinite(v) for v in logs.values()), logs
    assert m.global_step == 2 + 2 * 2                         # 2 pre-training steps, then G+D per batch
    assert "total_loss/discriminator" in logs and "gen_adv_loss/train_mel_loss" in logs
    assert not torch.equal(w0, m.generator.encoder.convnext[0].pwconv1_weight)
    assert not torch.equal(d0, m.discriminator.multiperioddisc.discriminators[0].convs[1].weight_v)
    # reference quirk: the decoder receives no gradient (segment.detach()); with zero Adam moments only weight decay
    # touches it, and at warm-up learning rates (<= 4e-7) that is below f32 resolution
    dec = m.generator.decoder.convnext[0].pwconv1_weight
    assert float(dec.grad.abs().max()) == 0.0 and relerr(dec, dec0) < 1e-6


def test_pipelined_steps_match_serial_steps():
    """``pipeline_steps`` (discriminator phase issued from its own stream, joined lazily) changes the schedule, not the
    result.  One step from identical weights must agree tightly (both phases, both optimizer updates); later steps are only
    checked for sanity, because the MAS alignment is discrete and amplifies atomics-order noise between ANY two runs."""
    from optispeech_amd import precision, rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    precision.set_precision("bf16")
    try:
        cfg = ModelConfig()
        batch = synthetic_batch(2, 24, 96, cfg, seed=5, device="cuda")
        r01 = torch.rand(2, generator=torch.Generator().manual_seed(1))
        out = []
        for pipe in (False, True):
            torch.manual_seed(3)
            rng.manual_seed(3, 0)
            rng._state["next_stream"] = 1                       # same dropout-site stream ids for both models
            m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
            m.pipeline_steps = pipe
            m.generator.segment_rand01 = r01
            m.optimizers()
            for sch in m.lr_schedulers():                       # no warm-up: the first update is a full-size AdamW step
                sch.warmup = 0
                sch.opt.lr = sch.base_lr
            m.training_step(batch, 0)
            logs = m.fetch_logs()                               # joins the discriminator stream
            sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
            m.training_step(batch, 1)
            m.training_step(batch, 2)
            logs3 = m.fetch_logs()
            assert all(np.isfinite(v) for v in logs3.values())
            out.append((sd, logs))
        (a, la), (b, lb) = out
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-3 * abs(la[k]) + 1e-5, (k, la[k], lb[k])
        moved = 0
        for k in a:
            # a first AdamW step moves every weight by +-lr (2e-4): a gradient whose sign is noise-level may flip -> 2 lr
            assert torch.allclose(a[k], b[k], rtol=1e-3, atol=5e-4), (k, (a[k] - b[k]).abs().max().item())
            moved += int("discriminator" in k and a[k].is_floating_point())
        assert moved > 0
    finally:
        precision.set_precision("f32")


@pytest.mark.parametrize("which", ["multiperioddisc", "multiresddisc"])
def test_f32_mode_split_bf16_stacks_match_torch_conv2d(which):
    """f32 parity mode: the hand-written stacks (split-bf16 products, ConvStackPreciseFn) vs torch's conv2d (MIOpen; test-side
    restatement in tests/_torch_disc_ref.py) on the same weights -- scores, feature maps, input gradient and every parameter gradient within 2e-4 of the tensor scale."""
    from optispeech_amd import precision
    from optispeech_amd.model import discriminator as D
    from oracle import schema as S
    precision.set_precision("f32")
    torch.manual_seed(0)
    m = (D.MultiPeriodDiscriminator() if which == "multiperioddisc" else D.MultiResolutionDiscriminator()).to("cuda")
    W = {k[len("discriminator." + which + "."):]: v for k, v in S.make_weights(S.discriminator_schema(), 4321).items() if which in k}
    m.load_state_dict(W)
    y = (torch.rand(3, 16384, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda()
    res = {}
    from tests import _torch_disc_ref as TR
    if True:
        for hip in (False, True):
            yh = (torch.rand(3, 16384, generator=torch.Generator().manual_seed(2)) * 2 - 1).cuda().requires_grad_(True)
            for p in m.parameters():
                p.requires_grad_(True)
                p.grad = None
            rs, gs, frs, fgs = m(y, yh) if hip else TR.multi(m, y, yh)
            loss = D._hinge_d(rs, gs) + 0.1 * D._feature_matching(frs, fgs) + D._hinge_g(gs)
            loss.backward()
            torch.cuda.synchronize()
            res[hip] = (loss.item(), [g.detach().clone() for g in gs], yh.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
    a, b = res[False], res[True]
    rel = lambda u, v: ((u.double() - v.double()).abs().max() / v.double().abs().max().clamp_min(1e-30)).item()   # noqa: E731
    assert abs(a[0] - b[0]) <= 1e-5 * abs(a[0])
    for u, v in zip(b[1], a[1]):                             # score maps: same values, different (channels-last) element order
        assert rel(u.flatten(1).sort(1).values, v.flatten(1).sort(1).values) < 2e-4
    # gradients pass through kinks (LeakyReLU, hinge clamp, |.|): an input whose pre-activation sits within round-off of a
    # kink flips a whole back-propagated path, so single elements may differ; compare in the L2 sense
    rel2 = lambda u, v: ((u.double() - v.double()).norm() / v.double().norm().clamp_min(1e-30)).item()   # noqa: E731
    assert rel2(b[2], a[2]) < 5e-3, rel2(b[2], a[2])
    for k in a[3]:
        if a[3][k].abs().max().item() > 1e-7:
            assert rel2(b[3][k], a[3][k]) < 5e-3, (k, rel2(b[3][k], a[3][k]))


def test_discriminator_replay_matches_second_forward():
    """``replay_disc_forward``: the discriminator phase on the generator phase's recorded forward gives the same loss and the
    same discriminator gradients as evaluating the stacks again."""
    from optispeech_amd import precision, rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    precision.set_precision("bf16")
    try:
        cfg = ModelConfig()
        batch = synthetic_batch(2, 24, 96, cfg, seed=5, device="cuda")
        r01 = torch.rand(2, generator=torch.Generator().manual_seed(1))
        out = []
        for rep in (False, True):
            torch.manual_seed(3)
            rng.manual_seed(3, 0)
            rng._state["next_stream"] = 1
            m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
            m.replay_disc_forward = rep
            m.generator.segment_rand01 = r01
            _, od = m.optimizers()
            od.step = lambda *a, **k: None                      # keep the gradient arena as the step left it
            m.training_step(batch, 0)
            logs = m.fetch_logs()
            torch.cuda.synchronize()
            out.append((logs["total_loss/discriminator"], od.arena.grad.clone()))
        (la, ga), (lb, gb) = out
        assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
        assert ga.abs().sum().item() > 0
        assert ((ga - gb).norm() / ga.norm()).item() < 1e-3
    finally:
        precision.set_precision("f32")
