#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING THE REFERENCE in the build container.

    python tools/make_golden.py            # writes tests/golden/*.npz

Needs /root/reference (read-only upstream tree); never runs on the GPU box.  Nothing from the
reference is copied: only inputs and the reference's numerical outputs are stored.  Weights are
not stored either -- the reference modules are instantiated and their parameters overwritten
with oracle.schema.make_weights(schema, seed), which the tests regenerate.

Stubs / shims:
  * omegaconf / hydra / lightning / rootutils / rich / matplotlib-free stand-ins: import-only.
  * numba.jit -> identity decorator.  Two numba-vs-python differences are shimmed:
      - _average_by_duration receives float length arrays (alignments.py:276-278); python slicing
        needs ints -> cast to int64.
      - alignments.py:188 `log_prob[0,:j+1].sum()`: numpy pairwise fp32 sum vs numba's sequential
        fp32 accumulation -> the search is run through a wrapper whose row 0 is
        np.cumsum(dtype=float32) (the semantics the build defines; *reasoned, not verified*).
        To keep goldens independent of that choice, every stored MAS case is also checked to
        give the identical path under the un-shimmed (pairwise) variant.
  * torchaudio is inert -> MelSpecReconstructionLoss is NOT exercised (parity unpinned).
"""
import functools
from collections import OrderedDict
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


# ----------------------------------------------------------------------------- stubs
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        return _Any()


def install_stubs():
    _stub("omegaconf", DictConfig=dict, OmegaConf=_Any(), open_dict=_Any())
    _stub("hydra", utils=_Any(), main=lambda *a, **k: (lambda f: f))
    _stub("hydra.core")
    _stub("hydra.core.hydra_config", HydraConfig=_Any())

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

    _stub("lightning", LightningModule=LightningModule, Callback=object, LightningDataModule=object, Trainer=object)
    _stub("lightning.pytorch")
    _stub("lightning.pytorch.loggers", Logger=object)
    _stub("lightning.pytorch.utilities", rank_zero_only=lambda f: f, grad_norm=lambda *a, **k: {})
    _stub("numba", jit=lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f)))
    _stub("torchaudio", transforms=_Any(), functional=_Any())
    for name in ("rootutils", "rich", "rich.syntax", "rich.tree", "rich.prompt", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name, **{"use": lambda *a, **k: None})
    sys.path.insert(0, REF)


install_stubs()
import optispeech.model.generator.alignments as RA                      # noqa: E402
from optispeech.model.generator import OptiSpeechGenerator              # noqa: E402
from optispeech.model.generator import modules as RM                    # noqa: E402
from optispeech.model.vocoder.wavenext import WaveNeXt                  # noqa: E402
from optispeech.model.vocoder.wavenext.disc import VocosDiscriminator   # noqa: E402
import optispeech.model.generator as RG                                 # noqa: E402
import optispeech.utils.segments as RSeg                                # noqa: E402

from oracle import schema as S                                          # noqa: E402

_orig_avg = RA._average_by_duration
RA._average_by_duration = lambda ds, xs, tl, fl: _orig_avg(ds, xs, tl.astype(np.int64), fl.astype(np.int64))
_orig_mas = RA._monotonic_alignment_search


def _mas_numba_semantics(lp):
    """Reference loop with row 0 replaced by a sequential fp32 prefix sum (see module docstring)."""
    T_mel, T_inp = lp.shape
    Q = np.full((T_inp, T_mel), fill_value=-np.inf)
    log_prob = lp.transpose(1, 0)
    Q[0, :] = np.cumsum(log_prob[0], dtype=np.float32)
    for j in range(1, T_mel):
        for i in range(1, min(j + 1, T_inp)):
            Q[i, j] = max(Q[i - 1, j - 1], Q[i, j - 1]) + log_prob[i, j]
    A = np.full((T_mel,), fill_value=T_inp - 1)
    for j in range(T_mel - 2, -1, -1):
        i_a, i_b = A[j + 1] - 1, A[j + 1]
        A[j] = 0 if i_b == 0 else (i_a if Q[i_a, j] >= Q[i_b, j] else i_b)
    return A


MAS_AGREE = []


def _mas_checked(lp):
    a = _mas_numba_semantics(lp)
    b = _orig_mas(lp)            # un-shimmed python/numpy-pairwise variant of the reference function
    MAS_AGREE.append(bool(np.array_equal(a, b)))
    return a


RA._monotonic_alignment_search = _mas_checked


# ----------------------------------------------------------------------------- model builders
def build_generator(c: S.Cfg, drop=0.0):
    P = functools.partial
    fe = SimpleNamespace(n_feats=c.n_feats, n_fft=c.n_fft, hop_length=c.hop, win_length=c.n_fft, sample_rate=22050,
                         f_min=80, f_max=8000)
    lc = SimpleNamespace(lambda_align=5.0, lambda_duration=1.0, lambda_pitch=1.0, lambda_energy=1.0)

    def pred(cls, spec, dr, **kw):
        return P(cls, num_layers=spec[0], intermediate_dim=spec[1], kernel_size=spec[2], dropout=dr,
                 conv_layer_class=torch.nn.Conv1d, **kw)

    g = OptiSpeechGenerator(
        dim=c.dim, segment_size=c.segment_size,
        text_embedding=P(RM.TextEmbedding, n_vocab=c.n_vocab, dropout=drop, padding_idx=0, max_source_positions=2000),
        encoder=P(RM.ConvNeXtBackbone, intermediate_dim=c.enc_inter, num_layers=c.enc_layers, drop_path=drop),
        duration_predictor=pred(RM.DurationPredictor, c.dur, drop),
        pitch_predictor=pred(RM.PitchPredictor, c.pitch, drop, embed_kernel_size=c.embed_kernel, embed_dropout=drop),
        energy_predictor=pred(RM.EnergyPredictor, c.energy, drop, embed_kernel_size=c.embed_kernel, embed_dropout=drop),
        decoder=P(RM.ConvNeXtBackbone, intermediate_dim=c.dec_inter, num_layers=c.dec_layers, drop_path=drop),
        vocoder=P(WaveNeXt, dim=c.voc_dim, intermediate_dim=c.voc_inter, num_layers=c.voc_layers, drop_path=drop),
        loss_coeffs=lc, feature_extractor=fe, num_speakers=1, num_languages=1, data_statistics=None)
    return g


def load_weights(module, weights, prefix):
    sd = module.state_dict()
    mine = {k[len(prefix):]: v for k, v in weights.items() if k.startswith(prefix)}
    persistent = {k: v for k, v in sd.items()}
    missing = set(persistent) - set(mine)
    extra = set(mine) - set(persistent)
    assert not extra, f"schema has keys the reference lacks: {sorted(extra)[:5]}"
    # the only reference keys the schema may omit are MR-STFT windows / torchaudio buffers
    assert all(("window" in k or "mel_spec" in k) for k in missing), sorted(missing)[:5]
    for k, v in mine.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    module.load_state_dict(mine, strict=False)


def make_batch(c, B, tt_rng, tm_rng, seed, wav=True):
    g = np.random.default_rng(seed)
    x_len = g.integers(tt_rng[0], tt_rng[1] + 1, B)
    m_len = g.integers(tm_rng[0], tm_rng[1] + 1, B)
    x_len[0], m_len[0] = tt_rng[1], tm_rng[1]
    Tt, Tm = int(x_len.max()), int(m_len.max())
    x = g.integers(1, 159, (B, Tt))
    for b in range(B):
        x[b, x_len[b]:] = 0
    mel = g.standard_normal((B, c.n_feats, Tm)).astype(np.float32)
    pit = g.standard_normal((B, Tm)).astype(np.float32)
    ene = g.standard_normal((B, Tm)).astype(np.float32)
    for b in range(B):
        mel[b, :, m_len[b]:] = 0
        pit[b, m_len[b]:] = 0
        ene[b, m_len[b]:] = 0
    out = dict(x=x.astype(np.int64), x_lengths=x_len.astype(np.int64), mel=mel, mel_lengths=m_len.astype(np.int64),
               pitches=pit, energies=ene)
    if wav:
        out["wav"] = g.uniform(-1, 1, (B, Tm * c.hop)).astype(np.float32)
    return out


def to_t(batch):
    return {k: torch.from_numpy(v) for k, v in batch.items()}


def run_generator_case(name, c, B, tt_rng, tm_rng, seed, with_disc, full_tensors=True, disc=None, dweights=None, gen=None, grad_probe=0):
    torch.manual_seed(0)
    if gen is None:
        gen = build_generator(c).train()                 # train mode, all dropout rates 0
        schema = S.generator_schema(c)
    else:                                                # another backbone: the schema IS the reference module's state dict
        gen = gen.train()
        schema = OrderedDict(("generator." + k, tuple(v.shape)) for k, v in gen.state_dict().items())
    weights = S.make_weights(schema, seed)
    load_weights(gen, weights, "generator.")
    batch = make_batch(c, B, tt_rng, tm_rng, seed + 1)
    tb = to_t(batch)
    rand01 = np.random.default_rng(seed + 2).uniform(0, 1, B).astype(np.float32)

    cap = {}
    # inject segment starts: same formula as utils/segments.py:29-34 with our uniform draws
    def _grs(x, x_lengths, segment_size):
        max_start = x_lengths - segment_size
        max_start[max_start < 0] = 0
        starts = (torch.from_numpy(rand01) * max_start).to(dtype=torch.long)
        return RSeg.get_segments(x, starts, segment_size), starts
    RG.get_random_segments = _grs
    _vd = RA.viterbi_decode

    def _viterbi(lp, tl, fl):
        ds, bl = _vd(lp, tl, fl)
        cap["durations"], cap["bin_loss"] = ds.detach().clone(), bl.detach().clone()
        return ds, bl
    RG.viterbi_decode = _viterbi
    _abd = RA.average_by_duration
    avg = []

    def _avg(*a):
        r = _abd(*a)
        avg.append(r.clone())
        return r
    RG.average_by_duration = _avg

    hooks = []
    def hook(nm):
        def f(mod, inp, out):
            cap[nm] = out
        return f
    for nm in ("text_embedding", "encoder", "alignment_module", "duration_predictor", "pitch_predictor",
               "energy_predictor", "feature_upsampler", "decoder", "vocoder", "forwardsum_loss"):
        hooks.append(getattr(gen, nm).register_forward_hook(hook(nm)))

    out = gen(x=tb["x"], x_lengths=tb["x_lengths"], mel=tb["mel"], mel_lengths=tb["mel_lengths"],
              pitches=tb["pitches"], energies=tb["energies"], sids=None, lids=None)
    for h in hooks:
        h.remove()
    res = dict(seed=np.int64(seed), rand01=rand01, **{"in_" + k: v for k, v in batch.items()})
    res["start_idx"] = out["start_idx"].numpy()
    res["loss"] = out["loss"].detach().numpy()
    for k in ("align_loss", "duration_loss", "pitch_loss", "energy_loss"):
        res[k] = out[k].numpy()
    res["bin_loss"] = cap["bin_loss"].numpy()
    res["forwardsum_loss"] = cap["forwardsum_loss"].detach().numpy()
    res["durations"] = cap["durations"].numpy()
    res["p_avg"], res["e_avg"] = avg[0].numpy(), avg[1].numpy()
    res["d_hat"] = cap["duration_predictor"].detach().numpy()
    res["p_hat"] = cap["pitch_predictor"][1].detach().numpy()
    res["e_hat"] = cap["energy_predictor"][1].detach().numpy()
    full = dict(text_emb=cap["text_embedding"][0], enc=cap["encoder"], log_p_attn=cap["alignment_module"],
                xp=cap["pitch_predictor"][0], xe=cap["energy_predictor"][0], y_up=cap["feature_upsampler"],
                dec=cap["decoder"], wav_hat=out["wav_hat"])
    for k, v in full.items():
        v = v.detach()
        if full_tensors:
            res[k] = v.numpy()
        else:  # checksums only
            fin = torch.where(torch.isfinite(v), v, torch.zeros_like(v))
            res[k + "_sum"] = fin.double().sum().numpy()
            res[k + "_l2"] = fin.double().norm().numpy()
    # --- G phase backward
    wav = torch.from_numpy(RSeg.get_segments_numpy(np.expand_dims(batch["wav"], 1), res["start_idx"] * c.hop,
                                                   out["segment_size"] * c.hop).squeeze(1))
    loss_g = out["loss"]
    if with_disc:
        for p in disc.parameters():
            p.requires_grad_(False)
        adv, logs, mr_term = _forward_gen_nomel(disc, wav, out["wav_hat"], want_mr=True)
        loss_g = loss_g + adv
        res["gen_adv_loss"] = adv.detach().numpy()
        for k, v in logs.items():
            res["genlog_" + k] = np.float64(v)
    gen.zero_grad()
    g_mr = None
    if with_disc and grad_probe:
        # the multi-resolution STFT term's own gradient (taken before the backward, graph retained): its log-magnitude part is
        # ill-conditioned at a random-init state (1 / |X| at near-empty bins), so that the element-wise probes of the VOCODER are
        # stored for the total gradient minus this term -- the adversarial + feature-matching gradient -- as well
        vp = [(k, p) for k, p in gen.named_parameters() if k.startswith("vocoder.")]
        g_mr = dict(zip([k for k, _ in vp], torch.autograd.grad(mr_term, [p for _, p in vp], retain_graph=True, allow_unused=True)))
    loss_g.backward()
    res["loss_g"] = loss_g.detach().numpy()
    gnames, gnorm, gnone = [], [], []
    for k, p in gen.named_parameters():
        if p.grad is None:
            gnone.append(k)
        else:
            gnames.append(k)
            gnorm.append(p.grad.double().norm().item())
            if full_tensors and p.numel() <= 70000:
                res["grad_g/" + k] = p.grad.numpy()
            elif grad_probe:                             # strided probe of <= grad_probe elements (tools/make_golden_b32.py)
                flat = p.grad.detach().reshape(-1)
                res["grad_g/" + k] = flat[:: max(1, flat.numel() // grad_probe)][:grad_probe].numpy().copy()
                res["gabs_g/" + k] = np.float64(flat.abs().max().item())
                if g_mr is not None and g_mr.get(k) is not None:
                    rest = (p.grad.detach() - g_mr[k]).reshape(-1)
                    res["grad_gns/" + k] = rest[:: max(1, rest.numel() // grad_probe)][:grad_probe].numpy().copy()
                    res["gabs_gns/" + k] = np.float64(rest.abs().max().item())
    res["grad_g_names"] = np.array(gnames)
    res["grad_g_norms"] = np.array(gnorm)
    res["grad_g_none"] = np.array(gnone)
    if with_disc:
        for p in disc.parameters():
            p.requires_grad_(True)
        disc.zero_grad()
        loss_d, dlog = disc.forward_disc(wav, out["wav_hat"].detach())
        loss_d.backward()
        res["loss_d"] = loss_d.detach().numpy()
        for k, v in dlog.items():
            res["disclog_" + k] = np.float64(v)
        dn, dv = [], []
        for k, p in disc.named_parameters():
            dn.append(k)
            dv.append(p.grad.double().norm().item())
            if grad_probe:
                flat = p.grad.detach().reshape(-1)
                res["grad_d/" + k] = flat[:: max(1, flat.numel() // grad_probe)][:grad_probe].numpy().copy()
                res["gabs_d/" + k] = np.float64(flat.abs().max().item())
            elif p.numel() <= 4096:
                res["grad_d/" + k] = p.grad.numpy()
        res["grad_d_names"] = np.array(dn)
        res["grad_d_norms"] = np.array(dv)
        res["wav"] = wav.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "loss", float(res["loss"]), "MAS shim==pairwise:", all(MAS_AGREE), len(MAS_AGREE))
    return gen, weights


def _forward_gen_nomel(disc, wav, wav_hat, want_mr=False):
    """VocosDiscriminator.forward_gen with the (inert torchaudio) mel term skipped."""
    _, g_mp, fr_mp, fg_mp = disc.multiperioddisc(y=wav, y_hat=wav_hat)
    _, g_mr, fr_mr, fg_mr = disc.multiresddisc(y=wav, y_hat=wav_hat)
    l_mp, ll_mp = disc.gen_loss(disc_outputs=g_mp)
    l_mr, ll_mr = disc.gen_loss(disc_outputs=g_mr)
    l_mp, l_mr = l_mp / len(ll_mp), l_mr / len(ll_mr)
    fm_mp = disc.feat_matching_loss(fmap_r=fr_mp, fmap_g=fg_mp) / len(fr_mp)
    fm_mr = disc.feat_matching_loss(fmap_r=fr_mr, fmap_g=fg_mr) / len(fr_mr)
    sc, mag = disc.mr_stft_loss(wav_hat, wav)
    mr = (sc + mag) * disc.lambda_mr_stft
    loss = l_mp + l_mr * disc.loss_coeffs.lambda_mrd + fm_mp + fm_mr * disc.loss_coeffs.lambda_mrd + mr
    logs = dict(loss_gen_mp=l_mp.item(), loss_gen_mrd=l_mr.item(), loss_fm_mp=fm_mp.item(),
                loss_fm_mrd=fm_mr.item(), mr_stft_loss=mr.item(), sc=sc.item(), mag=mag.item())
    return (loss, logs, mr) if want_mr else (loss, logs)


def build_disc(seed):
    fe = SimpleNamespace(n_feats=100, n_fft=1024, hop_length=256, win_length=1024, sample_rate=22050, f_min=80, f_max=8000)
    lc = SimpleNamespace(lambda_mrd=1.0, lambda_mel=45.0, lambda_mr_stft=2.5)
    d = VocosDiscriminator(feature_extractor=fe, loss_coeffs=lc)
    w = S.make_weights(S.discriminator_schema(), seed)
    load_weights(d, w, "discriminator.")
    return d, w


def run_synth_case(name, c, seed):
    gen = build_generator(c).eval()
    weights = S.make_weights(S.generator_schema(c), seed)
    load_weights(gen, weights, "generator.")
    g = np.random.default_rng(seed + 5)
    B = 3
    x_len = np.array([21, 13, 17])
    x = g.integers(1, 159, (B, 21))
    for b in range(B):
        x[b, x_len[b]:] = 0
    # random weights predict ~1-frame durations; bias the duration head so lengths are non-trivial
    with torch.no_grad():
        gen.duration_predictor.linear.bias.fill_(1.2)
    out = gen.synthesise(torch.from_numpy(x), torch.from_numpy(x_len), d_factor=1.1, p_factor=1.6, e_factor=1.2)
    res = dict(seed=np.int64(seed), in_x=x.astype(np.int64), in_x_lengths=x_len.astype(np.int64), dur_bias=np.float32(1.2),
               wav=out["wav"].numpy(), wav_lengths=out["wav_lengths"].numpy(), durations=out["durations"].numpy(),
               pitch=out["pitch"].numpy(), energy=out["energy"].numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "durations sum", res["durations"].sum(1), "wav", res["wav"].shape)


def run_unit_cases():
    g = np.random.default_rng(7)
    res = {}
    # MAS on random log-prob matrices incl. edge shapes (N=1, T==N, T>>N)
    shapes = [(5, 1), (6, 6), (40, 7), (97, 23), (300, 64)]
    for i, (T, N) in enumerate(shapes):
        lp = np.log(g.dirichlet(np.ones(N), size=T)).astype(np.float32)
        res[f"mas{i}_lp"] = lp
        res[f"mas{i}_path"] = _mas_checked(lp).astype(np.int64)
    # beta-binomial prior tables
    am = RA.AlignmentModule(adim=8, odim=4)
    tl, fl = torch.tensor([5, 3, 7]), torch.tensor([11, 9, 20])
    res["prior_tl"], res["prior_fl"] = tl.numpy(), fl.numpy()
    res["prior"] = am._generate_prior(tl, fl).numpy()
    # average_by_duration / expand_by_duration / gaussian upsampling
    ds = torch.tensor([[3., 0., 4., 2., 0.], [1., 5., 0., 0., 0.]])
    xs = torch.from_numpy(g.standard_normal((2, 9, 1)).astype(np.float32))
    res["abd_ds"], res["abd_xs"] = ds.numpy(), xs.numpy()
    res["abd_out"] = RA.average_by_duration(ds, xs, torch.tensor([5, 2]), torch.tensor([9, 6])).numpy()
    dur = torch.tensor([[2, 0, 3, 1], [1, 1, 0, 0]])
    xv = torch.from_numpy(g.standard_normal((2, 4, 3)).astype(np.float32))
    ex, ln = RA.expand_by_duration(xv, dur)
    res["exp_dur"], res["exp_x"], res["exp_out"], res["exp_len"] = dur.numpy(), xv.numpy(), ex.numpy(), ln.numpy()
    hs = torch.from_numpy(g.standard_normal((2, 4, 6)).astype(np.float32))
    hm = torch.tensor([[True] * 6, [True, True] + [False] * 4])
    dm = torch.tensor([[True] * 4, [True, True, False, False]])
    res["gu_hs"], res["gu_hm"], res["gu_dm"] = hs.numpy(), hm.numpy(), dm.numpy()
    res["gu_out"] = RA.GaussianUpsampling()(hs, dur.float(), hm, dm).numpy()
    # duration infer + regression losses + forward-sum loss
    from optispeech.model.generator.loss import FastSpeech2Loss, ForwardSumLoss
    d_o = torch.from_numpy(g.standard_normal((2, 5, 1)).astype(np.float32))
    p_o = torch.from_numpy((2 * g.standard_normal((2, 5, 1))).astype(np.float32))
    e_o = torch.from_numpy((2 * g.standard_normal((2, 5, 1))).astype(np.float32))
    ps = torch.from_numpy(g.standard_normal((2, 5, 1)).astype(np.float32))
    es = torch.from_numpy(g.standard_normal((2, 5, 1)).astype(np.float32))
    il = torch.tensor([5, 2])
    dl, pl, el = FastSpeech2Loss()(d_o, p_o, e_o, ds.unsqueeze(-1), ps, es, il)
    res.update(fs2_d=d_o.numpy(), fs2_p=p_o.numpy(), fs2_e=e_o.numpy(), fs2_ps=ps.numpy(), fs2_es=es.numpy(),
               fs2_il=il.numpy(), fs2_out=np.array([dl.item(), pl.item(), el.item()]))
    lp = torch.log_softmax(torch.from_numpy(g.standard_normal((2, 12, 5)).astype(np.float32)), -1)
    lp[1, :, 3:] = -np.inf
    lp[1, 9:] = -np.inf
    lp.requires_grad_(True)
    fsl = ForwardSumLoss()(lp, torch.tensor([5, 3]), torch.tensor([12, 9]))
    fsl.backward()
    res.update(fsl_lp=lp.detach().numpy(), fsl_out=fsl.detach().numpy(), fsl_grad=lp.grad.numpy())
    # MR-STFT loss + MRD spectrogram on random waves
    from optispeech.model.vocoder.wavenext.disc.loss import MultiResolutionSTFTLoss
    x = torch.from_numpy(g.uniform(-1, 1, (2, 4096)).astype(np.float32)).requires_grad_(True)
    y = torch.from_numpy(g.uniform(-1, 1, (2, 4096)).astype(np.float32))
    sc, mag = MultiResolutionSTFTLoss()(x, y)
    (sc + mag).backward()
    res.update(stft_x=x.detach().numpy(), stft_y=y.numpy(), stft_sc=sc.detach().numpy(), stft_mag=mag.detach().numpy(),
               stft_grad=x.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "units.npz"), **res)
    print("units ok; MAS shim==pairwise:", all(MAS_AGREE))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    run_unit_cases()
    disc, _ = build_disc(4321)
    run_generator_case("gen_small_am", S.SMALL, 3, (17, 24), (90, 120), 1234, with_disc=False)
    run_generator_case("gen_small_gan", S.SMALL, 2, (17, 24), (90, 120), 2345, with_disc=True, disc=disc)
    run_synth_case("synth_small", S.SMALL, 3456)
    run_generator_case("gen_full_b2", S.Cfg(), 2, (100, 128), (640, 800), 4567, with_disc=False, full_tensors=False)


if __name__ == "__main__":
    main()
