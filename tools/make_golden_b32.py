#!/usr/bin/env python3
"""Checksum fixtures at the BENCHMARK'S OWN SIZE, produced by RUNNING THE REFERENCE in the build container.

    python tools/make_golden_b32.py [gan] [synth] [transformer] [transformer_gan]     # writes tests/golden/full_b32_*.npz (a few hundred KB in all)

  * ``full_b32_gan``   BASELINE configs[1]: B = 32, T_text <= 128, T_mel <= 800, ConvNeXt generator + the GAN step (G phase with the
                       frozen discriminators, D phase), dropout rates 0, fixed segment starts.  Stored: integer paths (durations,
                       start indices: exact), the loss scalars, L2 / sum checksums of the big tensors, every gradient NORM and, per parameter, a strided
                       probe of <= 64 gradient elements (round 6).
  * ``full_b64_synth`` BASELINE configs[4]: 64 sentences, T_text in 64..128, the duration head biased to ~6 frames / phoneme
                       (random-init weights predict ~1 frame): int64 durations, wav lengths, per-sentence
                       waveform checksums.
  * ``full_b32_transformer`` BASELINE configs[3]: the reference ``Transformer`` encoder module at the full width (dim 256, 2 heads,
                       1 024 linear units, 4 blocks) on a ragged B = 32, T = 800 batch: output / input-gradient checksums, every
                       parameter-gradient norm.

Inputs are NOT stored (mel alone would be 10 MB): ``tests/_golden_inputs.py`` regenerates them from the seed with the same numpy
Generator calls as tools/make_golden.make_batch, and the fixture carries checksums of the inputs so that a drifted regeneration
fails loudly instead of comparing different problems.  Weights come from oracle.schema.make_weights(schema, seed) as everywhere.
Nothing of the reference's source is copied; it never runs on the GPU box.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.make_golden as MG                        # noqa: E402  (installs the import stubs, imports the reference)
from oracle import schema as S                        # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _cks(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.sqrt((a * a).sum())])


PROBE = 64            # gradient elements kept per parameter


def _grad_probes(g, res):
    """Per parameter a strided probe of <= 64 gradient ELEMENTS (a norm cannot see a sign flip or a permutation
    inside a tensor): flat[:: max(1, numel // 64)][:64], concatenated in the order of grad_{g,d}_names, f32, plus each tensor's
    max |gradient| (the scale an element-wise error is read against)."""
    for fam in ("g", "d", "gns"):                             # gns: generator gradient WITHOUT the MR-STFT term (vocoder parameters only)
        names = g["grad_%s_names" % ("g" if fam == "gns" else fam)].tolist()
        vals, lens, amax = [], [], []
        for k in names:
            key = "grad_%s/%s" % (fam, k)
            if key not in g.files:
                lens.append(0); amax.append(0.0)
                continue
            pr = np.asarray(g[key], dtype=np.float32).reshape(-1)      # already the strided probe (run_generator_case(grad_probe=PROBE))
            vals.append(pr); lens.append(pr.size); amax.append(float(g["gabs_%s/%s" % (fam, k)]))
        res["grad_%s_probe" % fam] = np.concatenate(vals) if vals else np.zeros(0, np.float32)
        res["grad_%s_probe_len" % fam] = np.array(lens, dtype=np.int64)
        res["grad_%s_absmax" % fam] = np.array(amax, dtype=np.float64)


def gan_case(name="full_b32_gan", B=32, seed=7788):
    t0 = time.time()
    disc, _ = MG.build_disc(seed + 11)
    tmp = "/tmp/osp_golden_b32"
    os.makedirs(tmp, exist_ok=True)
    keep_out, MG.OUT = MG.OUT, tmp
    try:
        MG.run_generator_case(name, S.Cfg(), B, (96, 128), (600, 800), seed, with_disc=True, full_tensors=False, disc=disc, grad_probe=PROBE)
    finally:
        MG.OUT = keep_out
    g = np.load(os.path.join(tmp, name + ".npz"), allow_pickle=False)
    res = {}
    for k in g.files:
        v = g[k]
        if k.startswith("in_"):
            if k in ("in_x_lengths", "in_mel_lengths"):
                res[k] = v
            res["cks_" + k] = _cks(v)                      # checksum of every regenerated input
            continue
        if k.startswith(("grad_d/", "grad_g/", "gabs_d/", "gabs_g/", "grad_gns/", "gabs_gns/")):
            continue                                       # norms + the strided probes (_grad_probes) at this size
        if k == "wav":
            res["wav_cks"] = _cks(v)
            continue
        res[k] = v
    _grad_probes(g, res)
    res["disc_seed"] = np.int64(seed + 11)
    res["batch_args"] = np.array([B, 96, 128, 600, 800, seed + 1], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "written in", round(time.time() - t0, 1), "s;", os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KB")


def transformer_gan_case(name="full_b32_transformer_gan", B=32, seed=6655):
    """BASELINE configs[3] as a WHOLE MODEL: the reference OptiSpeechGenerator with the Transformer encoder / decoder of
    configs/model/generator/{encoder,decoder}/transformer.yaml (dropout rates 0) + the GAN step, B = 32, T_text <= 128, T_mel <= 800.
    Weights: oracle.schema.make_weights over the reference module's own state-dict names and shapes (stored, so that the test can
    check its model exposes exactly those).  Checksums / norms only, like full_b32_gan."""
    import functools
    from types import SimpleNamespace
    from optispeech.model.generator import OptiSpeechGenerator
    from optispeech.model.generator import modules as RM
    from optispeech.model.vocoder.wavenext import WaveNeXt
    t0 = time.time()
    c = S.Cfg()
    P = functools.partial
    fe = SimpleNamespace(n_feats=c.n_feats, n_fft=c.n_fft, hop_length=c.hop, win_length=c.n_fft, sample_rate=22050, f_min=80, f_max=8000)
    lc = SimpleNamespace(lambda_align=5.0, lambda_duration=1.0, lambda_pitch=1.0, lambda_energy=1.0)
    tr = P(RM.Transformer, attention_heads=2, linear_units=1024, num_blocks=4, dropout_rate=0.0, positional_dropout_rate=0.0,
           attention_dropout_rate=0.0, normalize_before=True, concat_after=False, positionwise_layer_type="conv1d",
           positionwise_conv_kernel_size=1, use_scaled_pos_enc=True, init_alpha=1.0, init_type="xavier_uniform")

    def pred(cls, spec, **kw):
        return P(cls, num_layers=spec[0], intermediate_dim=spec[1], kernel_size=spec[2], dropout=0.0, conv_layer_class=torch.nn.Conv1d, **kw)
    gen = OptiSpeechGenerator(
        dim=c.dim, segment_size=c.segment_size,
        text_embedding=P(RM.TextEmbedding, n_vocab=c.n_vocab, dropout=0.0, padding_idx=0, max_source_positions=2000),
        encoder=tr, duration_predictor=pred(RM.DurationPredictor, c.dur),
        pitch_predictor=pred(RM.PitchPredictor, c.pitch, embed_kernel_size=c.embed_kernel, embed_dropout=0.0),
        energy_predictor=pred(RM.EnergyPredictor, c.energy, embed_kernel_size=c.embed_kernel, embed_dropout=0.0),
        decoder=tr, vocoder=P(WaveNeXt, dim=c.voc_dim, intermediate_dim=c.voc_inter, num_layers=c.voc_layers, drop_path=0.0),
        loss_coeffs=lc, feature_extractor=fe, num_speakers=1, num_languages=1, data_statistics=None)
    disc, _ = MG.build_disc(seed + 11)
    tmp = "/tmp/osp_golden_b32"
    os.makedirs(tmp, exist_ok=True)
    keep_out, MG.OUT = MG.OUT, tmp
    try:
        MG.run_generator_case(name, c, B, (96, 128), (600, 800), seed, with_disc=True, full_tensors=False, disc=disc, gen=gen, grad_probe=PROBE)
    finally:
        MG.OUT = keep_out
    g = np.load(os.path.join(tmp, name + ".npz"), allow_pickle=False)
    res = {}
    for k in g.files:
        v = g[k]
        if k.startswith("in_"):
            if k in ("in_x_lengths", "in_mel_lengths"):
                res[k] = v
            res["cks_" + k] = _cks(v)
            continue
        if k.startswith(("grad_d/", "grad_g/", "gabs_d/", "gabs_g/", "grad_gns/", "gabs_gns/")):
            continue
        if k == "wav":
            res["wav_cks"] = _cks(v)
            continue
        res[k] = v
    _grad_probes(g, res)
    sd = gen.state_dict()
    res["state_names"] = np.array(list(sd.keys()))
    res["state_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    res["disc_seed"] = np.int64(seed + 11)
    res["batch_args"] = np.array([B, 96, 128, 600, 800, seed + 1], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "written in", round(time.time() - t0, 1), "s;", os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KB")


def synth_case(name="full_b64_synth", B=64, seed=8899):
    t0 = time.time()
    c = S.Cfg()
    gen = MG.build_generator(c).eval()
    weights = S.make_weights(S.generator_schema(c), seed)
    MG.load_weights(gen, weights, "generator.")
    g = np.random.default_rng(seed + 5)
    x_len = g.integers(64, 129, B)
    x_len[0] = 128
    x = g.integers(1, 159, (B, 128))
    for b in range(B):
        x[b, x_len[b]:] = 0
    bias = float(np.log(6.25))
    with torch.no_grad():
        gen.duration_predictor.linear.bias.fill_(bias)
    out = gen.synthesise(torch.from_numpy(x), torch.from_numpy(x_len), d_factor=1.0, p_factor=1.6, e_factor=1.2)
    wav = out["wav"].numpy().astype(np.float64)
    wl = out["wav_lengths"].numpy()
    res = dict(seed=np.int64(seed), in_x=x.astype(np.int64), in_x_lengths=x_len.astype(np.int64), dur_bias=np.float32(bias),
               factors=np.array([1.0, 1.6, 1.2]), durations=out["durations"].numpy(), wav_lengths=wl,
               wav_shape=np.array(wav.shape, dtype=np.int64),
               wav_sum=np.array([wav[b, :wl[b]].sum() for b in range(B)]),
               wav_l2=np.array([np.sqrt((wav[b, :wl[b]] ** 2).sum()) for b in range(B)]),
               # strided samples of every waveform: an element-wise check that a checksum cannot fake (64 x 257 values)
               wav_probe=np.stack([wav[b, :: max(1, wav.shape[1] // 256)][:257] for b in range(B)]).astype(np.float32),
               wav_probe_step=np.int64(max(1, wav.shape[1] // 256)),
               pitch_cks=_cks(out["pitch"].numpy()), energy_cks=_cks(out["energy"].numpy()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "frames", int(res["durations"].sum()), "wav", wav.shape, "written in", round(time.time() - t0, 1), "s;",
          os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KB")


def transformer_case(name="full_b32_transformer", B=32, T=800, seed=9911):
    t0 = time.time()
    from optispeech.model.generator.modules.transformer import Transformer
    cfg = dict(attention_heads=2, linear_units=1024, num_blocks=4, dropout_rate=0.2, positional_dropout_rate=0.2,
               attention_dropout_rate=0.2, normalize_before=True, concat_after=False, positionwise_layer_type="conv1d",
               positionwise_conv_kernel_size=1, use_scaled_pos_enc=True, init_alpha=1.0, init_type="xavier_uniform")
    torch.manual_seed(seed)
    m = Transformer(dim=256, **cfg).eval()
    sd = m.state_dict()
    g = np.random.default_rng(seed)
    # weights: the init pattern plus a seeded perturbation regenerated by the test (biases are zero at init); stored as
    # (key order, shapes) + the seed -- not the 3.2 M values
    init = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        for k in sd:
            sd[k].copy_(torch.from_numpy(g.standard_normal(tuple(sd[k].shape)).astype(np.float32)) * (0.05 if sd[k].dim() > 1 else 0.02)
                        + (1.0 if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("after_norm.weight") else 0.0)
                        + (init[k] if k.endswith("alpha") else 0.0))
    m.load_state_dict(sd)
    lens = g.integers(600, T + 1, B)
    lens[0] = T
    x = torch.from_numpy(g.standard_normal((B, T, 256)).astype(np.float32)).requires_grad_(True)
    pad = torch.arange(T)[None] >= torch.from_numpy(lens)[:, None]
    y = m(x, pad)
    G = torch.from_numpy(g.standard_normal((B, T, 256)).astype(np.float32)) * (~pad)[:, :, None]
    (y * G).sum().backward()
    valid = (~pad)[:, :, None].numpy()
    res = dict(seed=np.int64(seed), lens=lens.astype(np.int64), keys=np.array(list(sd.keys())),
               shapes=np.array([",".join(map(str, v.shape)) for v in sd.values()]),
               y_cks=_cks(y.detach().numpy() * valid), dx_cks=_cks(x.grad.numpy() * valid),
               y_probe=(y.detach().numpy() * valid)[:, ::97, ::31].astype(np.float32),
               gnames=np.array([k for k, _ in m.named_parameters()]),
               gnorms=np.array([p.grad.double().norm().item() for _, p in m.named_parameters()]),
               w_cks=np.array([_cks(v.numpy()) for v in m.state_dict().values()]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "written in", round(time.time() - t0, 1), "s;", os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["gan", "synth", "transformer"]
    if "gan" in which:
        gan_case()
    if "synth" in which:
        synth_case()
    if "transformer" in which:
        transformer_case()
    if "transformer_gan" in which:
        transformer_gan_case()
