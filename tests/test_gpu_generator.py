"""Alignment / loss kernels and the whole generator (forward + backward) on the GPU vs goldens produced by the
reference and vs the CPU oracle.  Integer paths (MAS path, durations, start indices, wav lengths) must be exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import alignment as OA                  # noqa: E402  (checker only)
from oracle import generator as OG                  # noqa: E402
from oracle import losses as OL                     # noqa: E402
from oracle import nn_ops as ON                     # noqa: E402
from oracle import schema as S                      # noqa: E402

DEV = "cuda"


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    assert torch.equal(fa, fb), "non-finite pattern differs"
    if fa.sum() == 0:
        return 0.0
    return ((a[fa] - b[fb]).abs().max() / b[fb].abs().max().clamp_min(1e-30)).item()


def lens(*v):
    return torch.tensor(v, dtype=torch.int64, device=DEV)


# ------------------------------------------------------------------------------------------------ alignment kernels
def test_prior_matches_scipy(golden):
    from optispeech_amd import kernels as K
    g = golden("units")
    tl, fl = g["prior_tl"], g["prior_fl"]
    got = K.betabinom_prior(lens(*tl), lens(*fl), int(fl.max()), int(tl.max()))
    assert relerr(got, g["prior"]) < 2e-6
    tl2, fl2 = torch.tensor([128, 97]), torch.tensor([800, 611])
    want = OA.batched_prior(tl2, fl2)
    got = K.betabinom_prior(tl2.to(DEV), fl2.to(DEV), 800, 128)
    assert relerr(got, want) < 2e-6


def test_mas_golden_exact(golden):
    from optispeech_amd import kernels as K
    g = golden("units")
    i = 0
    while f"mas{i}_lp" in g.files:
        lp, want = g[f"mas{i}_lp"], g[f"mas{i}_path"]
        T, N = lp.shape
        path, dur, bin_item = K.mas(torch.from_numpy(lp)[None].contiguous().to(DEV), lens(N), lens(T))
        assert np.array_equal(path[0, :T].cpu().numpy(), want), i
        assert np.array_equal(dur[0].cpu().numpy(), np.bincount(want, minlength=N).astype(np.float32))
        assert abs(bin_item.item() + lp[np.arange(T), want].mean()) < 1e-5
        i += 1
    assert i >= 5


@pytest.mark.parametrize("B,Tm,Nm", [(8, 800, 128), (3, 257, 64), (2, 500, 300), (2, 90, 65), (1, 1500, 700)])
def test_mas_ragged_batches_vs_oracle(B, Tm, Nm):
    from optispeech_amd import kernels as K
    g = np.random.default_rng(B * 1000 + Nm)
    tl = g.integers(max(1, Nm // 2), Nm + 1, B); tl[0] = Nm
    fl = np.maximum(g.integers(Tm // 2, Tm + 1, B), tl); fl[0] = Tm
    lp = np.full((B, Tm, Nm), -np.inf, np.float32)
    for b in range(B):
        lp[b, :fl[b], :tl[b]] = np.log(g.dirichlet(np.ones(tl[b]), size=fl[b]))
    path, dur, bin_item = K.mas(torch.from_numpy(lp).to(DEV), lens(*tl), lens(*fl))
    for b in range(B):
        want = OA.mas_path_c(lp[b, :fl[b], :tl[b]])
        assert np.array_equal(path[b, :fl[b]].cpu().numpy(), want), b
        assert np.array_equal(dur[b].cpu().numpy(), np.bincount(want, minlength=Nm).astype(np.float32))


def test_duration_stats_and_upsampling(golden):
    from optispeech_amd import kernels as K
    from optispeech_amd import ops
    g = golden("units")
    ds = torch.from_numpy(g["abd_ds"]).to(DEV)
    xs = torch.from_numpy(g["abd_xs"][..., 0]).contiguous().to(DEV)
    a0, a1, _ = K.duration_stats(ds, xs, xs * 2, lens(5, 2), lens(9, 6))
    assert relerr(a0, g["abd_out"]) < 1e-6 and relerr(a1, 2 * g["abd_out"]) < 1e-6
    dur = torch.from_numpy(g["exp_dur"]).to(DEV)
    ex = K.expand_by_duration(torch.from_numpy(g["exp_x"]).to(DEV), dur, int(g["exp_len"].max()))
    assert np.array_equal(ex.cpu().numpy(), g["exp_out"])
    # gaussian upsampling: golden masks correspond to x_len = (4, 2), y_len = (6, 2)
    hs = torch.from_numpy(g["gu_hs"])
    hs8 = torch.cat([hs, hs[:, :, :2]], -1).contiguous()       # C must be a multiple of 4 for the GEMM fast path
    want = OA.gaussian_upsampling(hs8, torch.from_numpy(g["exp_dur"]).float(), torch.from_numpy(g["gu_hm"]),
                                  torch.from_numpy(g["gu_dm"]))
    got = ops.GaussianUpsampleFn.apply(hs8.to(DEV), dur.float(), lens(4, 2), lens(6, 2), 6, 0.1)
    assert relerr(got, want) < 1e-5
    assert relerr(got[..., :6], g["gu_out"]) < 1e-5


def test_losses_vs_golden(golden):
    from optispeech_amd import ops
    g = golden("units")
    d, p, e = (torch.from_numpy(g[k][..., 0]).to(DEV).requires_grad_(True) for k in ("fs2_d", "fs2_p", "fs2_e"))
    ds = torch.from_numpy(g["abd_ds"]).to(DEV)
    ps, es = torch.from_numpy(g["fs2_ps"][..., 0]).to(DEV), torch.from_numpy(g["fs2_es"][..., 0]).to(DEV)
    out = ops.VarianceLossFn.apply(d, p, e, ds, ps, es, lens(*g["fs2_il"]))
    assert relerr(torch.stack(out), g["fs2_out"]) < 1e-5
    (out[0] * 2 + out[1] * 3 + out[2] * 5).backward()
    dc, pc, ec = (torch.from_numpy(g[k][..., 0]).requires_grad_(True) for k in ("fs2_d", "fs2_p", "fs2_e"))
    oc = OL.variance_losses(dc, pc, ec, ds.cpu(), ps.cpu(), es.cpu(), torch.from_numpy(g["fs2_il"]))
    (oc[0] * 2 + oc[1] * 3 + oc[2] * 5).backward()
    assert relerr(d.grad, dc.grad) < 1e-5 and relerr(p.grad, pc.grad) < 1e-5 and relerr(e.grad, ec.grad) < 1e-5
    # forward-sum CTC: loss and gradient as produced by the reference (F.ctc_loss)
    from optispeech_amd import kernels as K
    lp = torch.from_numpy(g["fsl_lp"]).to(DEV)
    li, grad = K.forwardsum_ctc(lp, lens(5, 3), lens(12, 9))
    assert abs(li.sum().item() / 2 - float(g["fsl_out"])) < 1e-5 * abs(float(g["fsl_out"]))
    assert relerr(grad, g["fsl_grad"]) < 1e-4


def test_forwardsum_ctc_baseline_shape_vs_oracle():
    from optispeech_amd import kernels as K
    B, Tm, Nm = 4, 800, 128
    g = np.random.default_rng(5)
    tl, fl = np.array([128, 100, 96, 111]), np.array([800, 640, 700, 777])
    lp = np.full((B, Tm, Nm), -np.inf, np.float32)
    for b in range(B):
        lp[b, :fl[b], :tl[b]] = np.log(g.dirichlet(np.ones(tl[b]) * 0.3, size=fl[b]) + 1e-12) - 0.5
    lpc = torch.from_numpy(lp).requires_grad_(True)
    want = OL.forward_sum_loss(lpc, torch.from_numpy(tl), torch.from_numpy(fl))
    want.backward()
    li, grad = K.forwardsum_ctc(torch.from_numpy(lp).to(DEV), lens(*tl), lens(*fl))
    assert abs(li.sum().item() / B - want.item()) < 2e-5 * abs(want.item())
    # 800-step log-space recursions in f32 carry ~1e-3 relative noise on BOTH sides (the reference's F.ctc_loss
    # included); the arbiter is the same oracle evaluated in f64.
    lpd = torch.from_numpy(lp).double().requires_grad_(True)
    OL.forward_sum_loss(lpd, torch.from_numpy(tl), torch.from_numpy(fl)).backward()
    assert relerr(grad, lpd.grad) < 5e-3
    assert relerr(grad, lpc.grad) < 5e-3


def test_alignment_logprob_forward_backward_vs_oracle():
    from optispeech_amd import ops, kernels as K
    B, Tm, Nm, C = 3, 150, 40, 256
    gen = torch.Generator().manual_seed(3)
    f = torch.randn(B, Tm, C, generator=gen).requires_grad_(True)
    e = torch.randn(B, Nm, C, generator=gen).requires_grad_(True)
    tl, fl = torch.tensor([40, 33, 17]), torch.tensor([150, 120, 64])
    xpad = torch.arange(Nm)[None] >= tl[:, None]
    prior = OA.batched_prior(tl, fl, Tm, Nm)
    want = OA.pairwise_logprob(e, f, xpad, prior)
    dl = torch.randn(B, Tm, Nm, generator=gen)
    valid = (torch.arange(Tm)[None, :, None] < fl[:, None, None]) & (torch.arange(Nm)[None, None, :] < tl[:, None, None])
    dl = dl * valid
    torch.where(valid, want, torch.zeros(())).mul(dl).sum().backward()
    fg, eg = f.detach().to(DEV).requires_grad_(True), e.detach().to(DEV).requires_grad_(True)
    pr = K.betabinom_prior(tl.to(DEV), fl.to(DEV), Tm, Nm)
    got = ops.AlignLogProbFn.apply(fg, eg, pr, tl.to(DEV), fl.to(DEV))
    assert relerr(got, want) < 1e-5
    got.backward(dl.to(DEV))
    assert relerr(fg.grad, f.grad) < 1e-4 and relerr(eg.grad, e.grad) < 1e-4


def test_text_embedding_vs_oracle():
    from optispeech_amd.model.modules import TextEmbedding
    sch = {"te.embed_tokens.weight": (250, 256), "te.embed_positions.scale": (1,)}
    P = S.make_weights(sch, 5)
    for v in P.values():
        v.requires_grad_(True)
    tok = torch.randint(0, 159, (3, 77), generator=torch.Generator().manual_seed(1))
    tok[1, 50:] = 0
    want = ON.text_embedding(tok, P, "te.")
    dy = torch.randn(3, 77, 256, generator=torch.Generator().manual_seed(2))
    want.backward(dy)
    m = TextEmbedding(256, 250, 0.1).to(DEV).eval()
    m.load_state_dict({k[3:]: v.detach() for k, v in P.items()})
    got, _ = m(tok.to(DEV))
    assert relerr(got, want) < 1e-5
    got.backward(dy.to(DEV))
    gE = P["te.embed_tokens.weight"].grad.clone()
    gE[0] = 0                                                   # padding_idx row receives no gradient
    assert relerr(m.embed_tokens.weight.grad, gE) < 1e-5
    assert relerr(m.embed_positions.scale.grad, P["te.embed_positions.scale"].grad) < 1e-4


# ------------------------------------------------------------------------------------------------ whole generator
def _small_cfg():
    from optispeech_amd.config import ModelConfig
    c = S.SMALL
    return ModelConfig(dim=c.dim, enc_inter=c.enc_inter, dec_inter=c.dec_inter, dur=c.dur + (0.0,), pitch=c.pitch + (0.0,),
                       energy=c.energy + (0.0,), voc_dim=c.voc_dim, voc_inter=c.voc_inter, voc_layers=c.voc_layers).no_dropout()


def _gen_from_golden(g, cfg, scfg):
    from optispeech_amd.config import make_generator
    gen = make_generator(cfg).to(DEV).train()
    W = S.make_weights(S.generator_schema(scfg), int(g["seed"]))
    missing, unexpected = gen.load_state_dict({k[len("generator."):]: v for k, v in W.items()}, strict=True)
    sd = gen.state_dict()
    for k, v in W.items():
        assert torch.equal(sd[k[len("generator."):]].cpu(), v), k            # reference schema round trip
    gen.segment_rand01 = torch.from_numpy(g["rand01"])
    return gen


def _run_gen(gen, g):
    b = {k[3:]: torch.from_numpy(g[k]).to(DEV) for k in g.files if k.startswith("in_") and k != "in_wav"}
    return gen(b["x"], b["x_lengths"], b["mel"], b["mel_lengths"], b["pitches"], b["energies"], None, None)


def _ref_grad(gen, name_ref):
    """gradient of the parameter the reference calls `name_ref`, converted to the reference layout."""
    mod_name, leaf = name_ref.rsplit(".", 1)
    for pname, p in gen.named_parameters():
        mn, lf = pname.rsplit(".", 1)
        mod = gen.get_submodule(mn)
        key, _, to_ref = mod._ref(lf) if hasattr(mod, "_ref") else (lf, None, None)
        if mn + "." + key == name_ref or (mn + "." + key).replace("..", ".") == name_ref:
            if p.grad is None:
                return None
            return to_ref(p.grad) if to_ref else p.grad
    raise KeyError(name_ref)


@pytest.mark.parametrize("name", ["gen_small_am", "gen_full_b2"])
def test_generator_training_forward_backward_vs_golden(golden, name):
    g = golden(name)
    full = name == "gen_small_am"
    from optispeech_amd.config import ModelConfig
    cfg = _small_cfg() if full else ModelConfig().no_dropout()
    gen = _gen_from_golden(g, cfg, S.SMALL if full else S.Cfg())
    out = _run_gen(gen, g)
    aux = out["_aux"]
    assert np.array_equal(out["start_idx"].cpu().numpy(), g["start_idx"])            # index exact
    assert np.array_equal(aux["durations"].cpu().numpy(), g["durations"])            # MAS + bincount exact
    for k, kk in (("p_avg", "p_avg"), ("e_avg", "e_avg"), ("duration_hat", "d_hat"), ("pitch_hat", "p_hat"),
                  ("energy_hat", "e_hat")):
        assert relerr(aux[k], g[kk]) < 1e-3, k
    for k in ("loss", "align_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(out[k].item() - float(g[k])) <= 1e-4 * abs(float(g[k])), k
    assert abs(aux["bin_loss"].item() - float(g["bin_loss"])) <= 1e-4 * abs(float(g["bin_loss"]))
    assert abs(aux["forwardsum_loss"].item() - float(g["forwardsum_loss"])) <= 1e-4 * abs(float(g["forwardsum_loss"]))
    if full:
        assert relerr(aux["log_p_attn"], g["log_p_attn"]) < 1e-4
        assert relerr(aux["decoder_out"], g["dec"]) < 1e-3
        assert relerr(out["wav_hat"], g["wav_hat"]) < 1e-3                          # waveform within 1e-3 (north_star)
    else:
        v = aux["log_p_attn"]
        fin = torch.where(torch.isfinite(v), v, torch.zeros_like(v)).double()
        assert abs(fin.norm().item() - float(g["log_p_attn_l2"])) <= 1e-4 * float(g["log_p_attn_l2"])
        assert abs(out["wav_hat"].double().norm().item() - float(g["wav_hat_l2"])) <= 1e-3 * float(g["wav_hat_l2"])
    out["loss"].backward()
    none = set(g["grad_g_none"].tolist())
    for k, n in zip(g["grad_g_names"].tolist(), g["grad_g_norms"].tolist()):
        gr = _ref_grad(gen, k)
        assert gr is not None, k
        assert abs(gr.double().norm().item() - n) <= 2e-3 * max(n, 1e-6) + 1e-8, (k, gr.double().norm().item(), n)
    for k in none:                                                                    # decoder / vocoder / energy embed
        gr = _ref_grad(gen, k)
        assert gr is None or float(gr.abs().max()) == 0.0, k
    if full:
        for key in g.files:
            if key.startswith("grad_g/"):
                assert relerr(_ref_grad(gen, key[len("grad_g/"):]), g[key]) < 2e-3, key


def test_synthesise_vs_golden(golden):
    g = golden("synth_small")
    from optispeech_amd.config import make_generator
    gen = make_generator(_small_cfg()).to(DEV).eval()
    W = S.make_weights(S.generator_schema(S.SMALL), int(g["seed"]))
    W["generator.duration_predictor.linear.bias"].fill_(float(g["dur_bias"]))
    gen.load_state_dict({k[len("generator."):]: v for k, v in W.items()})
    out = gen.synthesise(torch.from_numpy(g["in_x"]).to(DEV), torch.from_numpy(g["in_x_lengths"]), d_factor=1.1,
                         p_factor=1.6, e_factor=1.2)
    assert np.array_equal(out["durations"].numpy(), g["durations"])                  # int64 exact
    assert np.array_equal(out["wav_lengths"].numpy(), g["wav_lengths"])
    assert relerr(out["pitch"], g["pitch"]) < 1e-3 and relerr(out["energy"], g["energy"]) < 1e-3
    assert relerr(out["wav"], g["wav"]) < 1e-3
    assert out["rtf"] > 0 and out["latency"] > 0


def test_multispeaker_sids_lids_vs_reference_golden(golden):
    """sids / lids -> sid_embed / lid_embed added to the encoder output (generator/__init__.py:62-65,112-117,235-246): the
    training forward + backward and synthesise() with explicit and with defaulted ids, against values the REFERENCE produced
    (tools/make_golden_multispeaker.py)."""
    import copy
    from optispeech_amd.config import make_generator
    g = golden("gen_small_multispk")
    cfg = copy.deepcopy(_small_cfg())
    cfg.num_speakers, cfg.num_languages = 3, 2
    schema = S.generator_schema(S.SMALL)
    schema["generator.sid_embed.weight"] = (3, S.SMALL.dim)
    schema["generator.lid_embed.weight"] = (2, S.SMALL.dim)
    W = S.make_weights(schema, int(g["seed"]))
    gen = make_generator(cfg).to(DEV).train()
    gen.load_state_dict({k[len("generator."):]: v for k, v in W.items()}, strict=True)
    gen.segment_rand01 = torch.from_numpy(g["rand01"])
    b = {k[3:]: torch.from_numpy(g[k]).to(DEV) for k in g.files if k.startswith("in_")}
    sids, lids = torch.from_numpy(g["sids"]).to(DEV), torch.from_numpy(g["lids"]).to(DEV)
    out = gen(b["x"], b["x_lengths"], b["mel"], b["mel_lengths"], b["pitches"], b["energies"], sids, lids)
    assert np.array_equal(out["_aux"]["durations"].cpu().numpy(), g["durations"])
    assert np.array_equal(out["start_idx"].cpu().numpy(), g["start_idx"])
    for k in ("loss", "align_loss", "duration_loss", "pitch_loss", "energy_loss"):
        assert abs(out[k].item() - float(g[k])) <= 1e-4 * abs(float(g[k])), k
    assert relerr(out["wav_hat"], g["wav_hat"]) < 1e-3
    out["loss"].backward()
    assert relerr(gen.sid_embed.weight.grad, g["grad_sid_embed"]) < 2e-3
    assert relerr(gen.lid_embed.weight.grad, g["grad_lid_embed"]) < 2e-3
    for k, n in zip(g["grad_g_names"].tolist(), g["grad_g_norms"].tolist()):
        got = _ref_grad(gen, k)
        assert got is not None and abs(got.double().norm().item() - n) <= 2e-3 * max(n, 1e-6) + 1e-8, k
    # inference
    gen.eval()
    with torch.no_grad():
        gen.duration_predictor.linear.bias.fill_(float(g["dur_bias"]))
    x, xl = torch.from_numpy(g["syn_x"]).to(DEV), torch.from_numpy(g["syn_x_lengths"])
    for tag, kw in (("syn", dict(sids=sids, lids=lids)), ("syn0", dict())):
        o = gen.synthesise(x, xl, d_factor=1.1, p_factor=1.6, e_factor=1.2, **kw)
        assert np.array_equal(o["durations"].numpy(), g[tag + "_durations"]), tag
        assert np.array_equal(o["wav_lengths"].numpy(), g[tag + "_wav_lengths"])
        assert relerr(o["wav"], g[tag + "_wav"]) < 1e-3, tag
