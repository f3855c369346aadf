"""CPU restatement of the Transformer backbone variant -- TEST INFRASTRUCTURE ONLY.

Follows (eval mode / dropout off; the dropout sites are listed so the HIP path can place its Philox streams identically):
  * Transformer.forward                 generator/modules/transformer.py:24-27        mask = ~padding_mask
  * Encoder.forward                     _transformer/encoder.py:266-313               embed -> encoders -> after_norm
  * ScaledPositionalEncoding.forward    _transformer/embedding.py:112-124             x + alpha * pe   (dropout)
  * EncoderLayer.forward (pre-LN)       _transformer/encoder_layer.py:60-116          x + drop(MHA(LN x)); x + drop(FFN(LN x))
  * MultiHeadedAttention                _transformer/attention.py:50-125              masked_fill(min) -> softmax -> masked_fill(0) (dropout)
  * MultiLayeredConv1d (k = 1)          _transformer/multi_layer_conv.py:47-62        w_2(drop(relu(w_1 x)))
  * LayerNorm eps 1e-12                 _transformer/layer_norm.py:20-23
Pinned by tests/golden/transformer.npz (tools/make_golden_transformer.py runs the reference module).
"""
import math

import torch
import torch.nn.functional as F


def sinusoid_pe(T, C):
    """_transformer/embedding.py:58-72 (interleaved sin / cos)."""
    pe = torch.zeros(T, C)
    pos = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, C, 2, dtype=torch.float32) * -(math.log(10000.0) / C))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def forward(P, x, padding_mask, heads, prefix="transformer."):
    """P: state dict (reference keys) -> (B, T, C)."""
    B, T, C = x.shape
    dk = C // heads
    keep = ~padding_mask                                         # (B, T) valid keys
    h = x + P[prefix + "embed.0.alpha"] * sinusoid_pe(T, C).to(x.dtype)
    n = 0
    while f"{prefix}encoders.{n}.norm1.weight" in P:
        n += 1
    for i in range(n):
        p = f"{prefix}encoders.{i}."
        z = F.layer_norm(h, (C,), P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-12)
        q = F.linear(z, P[p + "self_attn.linear_q.weight"], P[p + "self_attn.linear_q.bias"]).view(B, T, heads, dk).transpose(1, 2)
        k = F.linear(z, P[p + "self_attn.linear_k.weight"], P[p + "self_attn.linear_k.bias"]).view(B, T, heads, dk).transpose(1, 2)
        v = F.linear(z, P[p + "self_attn.linear_v.weight"], P[p + "self_attn.linear_v.bias"]).view(B, T, heads, dk).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
        m = ~keep[:, None, None, :]
        a = torch.softmax(s.masked_fill(m, torch.finfo(s.dtype).min), dim=-1).masked_fill(m, 0.0)
        o = torch.matmul(a, v).transpose(1, 2).contiguous().view(B, T, C)
        h = h + F.linear(o, P[p + "self_attn.linear_out.weight"], P[p + "self_attn.linear_out.bias"])
        z = F.layer_norm(h, (C,), P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-12)
        f = torch.relu(F.linear(z, P[p + "feed_forward.w_1.weight"][:, :, 0], P[p + "feed_forward.w_1.bias"]))
        h = h + F.linear(f, P[p + "feed_forward.w_2.weight"][:, :, 0], P[p + "feed_forward.w_2.bias"])
    return F.layer_norm(h, (C,), P[prefix + "after_norm.weight"], P[prefix + "after_norm.bias"], 1e-12)
