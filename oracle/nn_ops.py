"""Oracle (test infrastructure): module-level restatements, channels-last, functional.

Parameters come from a flat dict ``P`` in the reference state-dict schema; ``pre``
is the key prefix of the module being evaluated.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- helpers
def length_mask(lengths, max_len=None):
    """True where t < length.  utils/model.py:12-16 (sequence_mask)."""
    if max_len is None:
        max_len = int(lengths.max())
    return torch.arange(max_len, device=lengths.device)[None, :] < lengths[:, None]


def conv1d_cl(x, w, b, pad):
    """Dense Conv1d on channels-last input. x (B,T,Cin); w (Cout,Cin,k) (torch schema)."""
    return F.conv1d(x.transpose(1, 2), w, b, padding=pad).transpose(1, 2)


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# --------------------------------------------------------------------------- ConvNeXt
def convnext_block(x, P, pre, row_scale=None):
    """ConvNeXtBlock.forward, generator/modules/convnext.py:34-47.

    x (B,T,C).  ``row_scale`` (B,) is the DropPath factor (bernoulli/keep_prob,
    convnext.py:121-129); None == eval / rate 0.
    """
    C = x.shape[-1]
    h = F.conv1d(x.transpose(1, 2), P[pre + "dwconv.weight"], P[pre + "dwconv.bias"],
                 padding=3, groups=C).transpose(1, 2)                       # :36
    h = layer_norm(h, P[pre + "norm.weight"], P[pre + "norm.bias"], 1e-6)   # :38
    h = F.linear(h, P[pre + "pwconv1.weight"], P[pre + "pwconv1.bias"])    # :39
    h = F.gelu(h)                                                           # :40 exact erf GELU
    h = F.linear(h, P[pre + "pwconv2.weight"], P[pre + "pwconv2.bias"])    # :41
    h = P[pre + "gamma"] * h                                                # :42-43
    if row_scale is not None:
        h = h * row_scale[:, None, None]
    return x + h                                                            # :46


def convnext_backbone(x, P, pre, padding_mask=None, row_scales=None, return_blocks=False):
    """ConvNeXtBackbone.forward, convnext.py:92-103.  x (B,T,C); padding_mask (B,T) True=pad."""
    n_layers = 0
    while (pre + f"convnext.{n_layers}.gamma") in P:
        n_layers += 1
    keep = None if padding_mask is None else (1.0 - padding_mask.float())[:, :, None]
    blocks = []
    for i in range(n_layers):
        rs = None if row_scales is None else row_scales[i]
        x = convnext_block(x, P, pre + f"convnext.{i}.", rs)
        if keep is not None:
            x = x * keep                                                    # :99-101
        blocks.append(x)
    y = layer_norm(x, P[pre + "final_layer_norm.weight"], P[pre + "final_layer_norm.bias"], 1e-6)
    return (y, blocks) if return_blocks else y


# --------------------------------------------------------------------------- text embedding
def sinusoid_table(T, dim, theta, device=None):
    """ScaledSinusoidalEmbedding buffers+forward (without the scale), modules/layers.py:48-71."""
    half = dim // 2
    inv_freq = theta ** -(torch.arange(half, device=device).float() / half)   # :54-56
    ang = torch.arange(T, device=device).float()[:, None] * inv_freq[None, :]   # :69
    return torch.cat((ang.sin(), ang.cos()), dim=-1)                           # :70


def text_embedding(tokens, P, pre, theta=2000, drop_mask=None):
    """TextEmbedding.forward, modules/core.py:25-31 (dropout given as an explicit keep/scale mask)."""
    W = P[pre + "embed_tokens.weight"]
    dim = W.shape[1]
    emb = math.sqrt(dim) * F.embedding(tokens, W, padding_idx=0)            # :27
    pos = sinusoid_table(tokens.shape[1], dim, theta, tokens.device) * P[pre + "embed_positions.scale"]  # layers.py:71
    x = emb + pos[None]                                                      # :29
    if drop_mask is not None:
        x = x * drop_mask
    return x


# --------------------------------------------------------------------------- variance predictors
def variance_predictor(x, padding_mask, P, pre, drop_masks=None):
    """VariancePredictor.forward, modules/core.py:83-97; LayerNorm(dim=1) eps 1e-12 layers.py:26-45."""
    i = 0
    while (pre + f"conv.{i}.0.weight") in P:
        w = P[pre + f"conv.{i}.0.weight"]
        k = w.shape[-1]
        x = conv1d_cl(x, w, P[pre + f"conv.{i}.0.bias"], (k - 1) // 2)      # core.py:66-71
        x = F.relu(x)                                                        # :72
        x = layer_norm(x, P[pre + f"conv.{i}.2.weight"], P[pre + f"conv.{i}.2.bias"], 1e-12)  # :73
        if drop_masks is not None:
            x = x * drop_masks[i]                                            # :74
        i += 1
    y = F.linear(x, P[pre + "linear.weight"], P[pre + "linear.bias"]).squeeze(-1)   # :95
    return y.masked_fill(padding_mask, 0.0)                                  # :96


def duration_infer(log_d, padding_mask, factor=1.0, clip_val=1e-8):
    """DurationPredictor.infer tail, modules/core.py:126-132. Returns int64."""
    d = torch.exp(log_d) - clip_val
    d = torch.ceil(d * factor)
    d = torch.clamp(d.long(), min=0)
    return d.masked_fill(padding_mask, 0)


def variance_embed_add(x, values, padding_mask, P, pre, drop_mask=None):
    """PitchPredictor.forward/.infer tail, modules/core.py:163-166 / :172-175.

    values (B,T) are the teacher-forced targets (train) or scaled predictions (infer).
    """
    w = P[pre + "embed.0.weight"]                                            # (dim,1,k)
    k = w.shape[-1]
    emb = conv1d_cl(values[:, :, None], w, P[pre + "embed.0.bias"], (k - 1) // 2)
    if drop_mask is not None:
        emb = emb * drop_mask
    x = x + emb
    return x * (1.0 - padding_mask.float())[..., None]


# --------------------------------------------------------------------------- WaveNeXt vocoder
def wavenext(x, P, pre, padding_mask=None, row_scales=None):
    """WaveNeXt.forward + WaveNeXtHead.forward, vocoder/wavenext/__init__.py:82-86, :31-48.

    x (B,T,Cin) channels-last (the reference passes (B,Cin,T)); returns (B, T*hop).
    """
    h = conv1d_cl(x, P[pre + "embed.weight"], P[pre + "embed.bias"], 3)      # :83
    h = layer_norm(h, P[pre + "norm.weight"], P[pre + "norm.bias"], 1e-6)    # :84
    h = convnext_backbone(h, P, pre + "backbone.", padding_mask, row_scales)  # :85
    h = F.linear(h, P[pre + "head.linear_1.weight"], P[pre + "head.linear_1.bias"])   # :43
    h = F.linear(h, P[pre + "head.linear_2.weight"])                         # :44
    audio = h.reshape(h.shape[0], -1)                                        # :45
    return torch.clip(audio, min=-1.0, max=1.0)                              # :47
