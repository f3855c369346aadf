"""Oracle (test infrastructure): reference state-dict schema + deterministic test weights.

``generator_schema`` / ``discriminator_schema`` list every persistent tensor of the reference model
(name -> shape) from the configuration numbers alone.
``tools/make_golden.py`` asserts that these agree with the real reference ``state_dict()`` and then
overwrites the reference weights with ``make_weights`` -- so goldens need not store weights: tests
regenerate the identical tensors from (schema, seed).
"""
from collections import OrderedDict
from dataclasses import dataclass, field

import numpy as np
import torch


@dataclass
class Cfg:
    dim: int = 256
    n_vocab: int = 250
    n_feats: int = 100
    n_fft: int = 1024
    hop: int = 256
    enc_layers: int = 4
    enc_inter: int = 1024
    dec_layers: int = 4
    dec_inter: int = 1024
    dur: tuple = (2, 384, 3)          # layers, channels, kernel
    pitch: tuple = (5, 256, 5)
    energy: tuple = (2, 384, 3)
    embed_kernel: int = 9
    voc_dim: int = 384
    voc_inter: int = 1152
    voc_layers: int = 8
    segment_size: int = 64


SMALL = Cfg(dim=64, enc_inter=128, dec_inter=128, dur=(2, 48, 3), pitch=(5, 64, 5), energy=(2, 48, 3),
            voc_dim=96, voc_inter=160, voc_layers=8)


def _convnext(s, pre, dim, inter, layers):
    for i in range(layers):
        p = f"{pre}convnext.{i}."
        s[p + "gamma"] = (dim,)
        s[p + "dwconv.weight"] = (dim, 1, 7)
        s[p + "dwconv.bias"] = (dim,)
        s[p + "norm.weight"] = (dim,)
        s[p + "norm.bias"] = (dim,)
        s[p + "pwconv1.weight"] = (inter, dim)
        s[p + "pwconv1.bias"] = (inter,)
        s[p + "pwconv2.weight"] = (dim, inter)
        s[p + "pwconv2.bias"] = (dim,)
    s[pre + "final_layer_norm.weight"] = (dim,)
    s[pre + "final_layer_norm.bias"] = (dim,)


def _predictor(s, pre, dim, spec):
    layers, ch, k = spec
    for i in range(layers):
        s[f"{pre}conv.{i}.0.weight"] = (ch, dim if i == 0 else ch, k)
        s[f"{pre}conv.{i}.0.bias"] = (ch,)
        s[f"{pre}conv.{i}.2.weight"] = (ch,)
        s[f"{pre}conv.{i}.2.bias"] = (ch,)
    s[pre + "linear.weight"] = (1, ch)
    s[pre + "linear.bias"] = (1,)


def generator_schema(c: Cfg, pre="generator."):
    s = OrderedDict()
    s[pre + "text_embedding.embed_tokens.weight"] = (c.n_vocab, c.dim)
    s[pre + "text_embedding.embed_positions.scale"] = (1,)
    _convnext(s, pre + "encoder.", c.dim, c.enc_inter, c.enc_layers)
    _predictor(s, pre + "duration_predictor.", c.dim, c.dur)
    a = pre + "alignment_module."
    for name, shape in (("t_conv1", (c.dim, c.dim, 3)), ("t_conv2", (c.dim, c.dim, 1)),
                        ("f_conv1", (c.dim, c.n_feats, 3)), ("f_conv2", (c.dim, c.dim, 3)),
                        ("f_conv3", (c.dim, c.dim, 1))):
        s[a + name + ".weight"] = shape
        s[a + name + ".bias"] = (shape[0],)
    for nm, spec in (("pitch_predictor.", c.pitch), ("energy_predictor.", c.energy)):
        _predictor(s, pre + nm + "predictor.", c.dim, spec)
        s[pre + nm + "embed.0.weight"] = (c.dim, 1, c.embed_kernel)
        s[pre + nm + "embed.0.bias"] = (c.dim,)
    _convnext(s, pre + "decoder.", c.dim, c.dec_inter, c.dec_layers)
    v = pre + "vocoder."
    s[v + "embed.weight"] = (c.voc_dim, c.dim, 7)
    s[v + "embed.bias"] = (c.voc_dim,)
    s[v + "norm.weight"] = (c.voc_dim,)
    s[v + "norm.bias"] = (c.voc_dim,)
    _convnext(s, v + "backbone.", c.voc_dim, c.voc_inter, c.voc_layers)
    s[v + "head.linear_1.weight"] = (c.n_fft + 2, c.voc_dim)
    s[v + "head.linear_1.bias"] = (c.n_fft + 2,)
    s[v + "head.linear_2.weight"] = (c.hop, c.n_fft + 2)
    return s


def discriminator_schema(pre="discriminator."):
    """MPD/MRD parameters (weight_norm g/v pairs) + MR-STFT window buffers. Mel-spec buffers of the
    torchaudio transform are not part of the oracle (parity unpinned)."""
    s = OrderedDict()
    chans = [(32, 1), (128, 32), (512, 128), (1024, 512), (1024, 1024)]
    for d in range(5):
        p = f"{pre}multiperioddisc.discriminators.{d}."
        for i, (co, ci) in enumerate(chans):
            s[p + f"convs.{i}.bias"] = (co,)
            s[p + f"convs.{i}.weight_g"] = (co, 1, 1, 1)
            s[p + f"convs.{i}.weight_v"] = (co, ci, 5, 1)
        s[p + "conv_post.bias"] = (1,)
        s[p + "conv_post.weight_g"] = (1, 1, 1, 1)
        s[p + "conv_post.weight_v"] = (1, 1024, 3, 1)
    ks = [(7, 5), (5, 3), (5, 3), (3, 3), (3, 3)]
    for d in range(3):
        p = f"{pre}multiresddisc.discriminators.{d}."
        for i, k in enumerate(ks):
            ci = 1 if i == 0 else 64
            s[p + f"convs.{i}.bias"] = (64,)
            s[p + f"convs.{i}.weight_g"] = (64, 1, 1, 1)
            s[p + f"convs.{i}.weight_v"] = (64, ci) + k
        s[p + "conv_post.bias"] = (1,)
        s[p + "conv_post.weight_g"] = (1, 1, 1, 1)
        s[p + "conv_post.weight_v"] = (1, 64, 3, 3)
    return s


def make_weights(schema, seed=1234, dtype=torch.float32):
    """Deterministic pseudo-random tensors for every schema entry (numpy PCG64, key-order independent)."""
    out = OrderedDict()
    for i, (name, shape) in enumerate(schema.items()):
        h = (seed * 1000003 + sum((j + 1) * ord(ch) for j, ch in enumerate(name))) % (2 ** 32)
        g = np.random.default_rng(h)
        n = g.standard_normal(shape).astype(np.float32)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "weight_g":
            a = 0.6 + 0.2 * np.abs(n)
        elif leaf == "gamma":
            a = 0.25 + 0.05 * n
        elif leaf == "scale":
            a = np.full(shape, 0.0625, np.float32) + 0.01 * n
        elif leaf == "bias":
            a = 0.05 * n
        elif len(shape) == 1:          # LayerNorm weights
            a = 1.0 + 0.1 * n
        elif "embed_tokens" in name:
            a = 0.3 * n
            a[0] = 0.0                 # padding_idx row
        else:
            fan_in = int(np.prod(shape[1:]))
            a = n * (1.0 / np.sqrt(fan_in))
        out[name] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(tuple(shape)).to(dtype)   # (0-d entries stay 0-d)
    return out
