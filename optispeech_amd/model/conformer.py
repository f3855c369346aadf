"""Host-side mirror of the Conformer backbone: ``optispeech/model/generator/modules/conformer.py``
wrapping the ESPnet-style encoder (``_conformer/encoder.py``, ``encoder_layer.py``, ``convolution.py``, ``swish.py``;
``_transformer/attention.py:208-313`` RelPositionMultiHeadedAttention, ``_transformer/embedding.py:252-322``
RelPositionalEncoding), in the configuration of configs/model/generator/{encoder,decoder}/conformer.yaml: rel_pos / rel_selfattn,
macaron feed-forward pair (conv1d k = 1), convolution module (k = 7 encoder / 31 decoder), swish, pre-LayerNorm.

Same class names and state-dict keys (``conformer.encoders.N.{self_attn.{linear_q,k,v,out,linear_pos,pos_bias_u,pos_bias_v},
feed_forward(_macaron).w_{1,2}, conv_module.{pointwise_conv1,depthwise_conv,norm,pointwise_conv2}, norm_*}``,
``conformer.after_norm``).  Arithmetic: every linear / pointwise conv, both attention products and the relative-position
product run on the conv-GEMM kernels (batched), masked softmax + attention dropout on osp_attn_softmax_*, the depthwise conv on
csrc/dwconv.hip, LayerNorms on osp_layernorm_*, dropout on osp_dropout_add; the rel-shift re-indexing, GLU, swish, BatchNorm
statistics and the residual sums are element-wise torch glue.
"""
import math

import torch
from torch import nn

from .. import ops, rng
from .base import RefSchemaModule, conv_to_native, conv_to_ref
from .lightspeech import _dw_to_native, _dw_to_ref
from .modules import FinalNorm
from .transformer import MultiLayeredConv1d, _Conv1dK1, _Linear, _xavier


class _LinearNoBias(RefSchemaModule):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(_xavier(cout, cin))

    def forward(self, x):
        return ops.conv_linear(x, self.weight, None, self.weight.shape[0], 1, 0, None)


def rel_positions(T, d_model, device):
    """RelPositionalEncoding.pe slice for a length-T input (embedding.py:275-321): (2T - 1, d_model), row i = encoding of relative
    position T - 1 - i (positive = key to the left)."""
    pos = torch.arange(T - 1, -T, -1.0, dtype=torch.float32).unsqueeze(1)                       # T-1 ... -(T-1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(2 * T - 1, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(device)


class RelPositionMultiHeadedAttention(nn.Module):
    """_transformer/attention.py:208-313 (zero_triu False)."""

    def __init__(self, n_head, n_feat, dropout_rate, zero_triu=False):
        super().__init__()
        assert n_feat % n_head == 0 and not zero_triu
        self.d_k, self.h, self.dropout_rate = n_feat // n_head, n_head, dropout_rate
        self.linear_q, self.linear_k = _Linear(n_feat, n_feat), _Linear(n_feat, n_feat)
        self.linear_v, self.linear_out = _Linear(n_feat, n_feat), _Linear(n_feat, n_feat)
        self.linear_pos = _LinearNoBias(n_feat, n_feat)
        self.pos_bias_u = nn.Parameter(_xavier(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(_xavier(self.h, self.d_k))
        self._stream = rng.new_stream()

    def forward(self, x, pos_emb, klen):
        B, T, C = x.shape
        H, dk = self.h, self.d_k
        q, k, v = self.linear_q(x), self.linear_k(x), self.linear_v(x)
        p = self.linear_pos(pos_emb.unsqueeze(0))[0]                                             # (2T-1, C)
        qu = q + self.pos_bias_u.reshape(1, 1, C)                                                # (q + u): heads are contiguous dk-slices of C
        qv = (q + self.pos_bias_v.reshape(1, 1, C)).view(B, T, H, dk).permute(2, 0, 1, 3).reshape(H, B * T, dk)
        ph = p.view(2 * T - 1, H, dk).permute(1, 0, 2)                                           # (H, 2T-1, dk)
        bd = ops.BatchedNTFn.apply(qv, ph)                                                       # (H, B*T, 2T-1)
        bd = bd.view(H, B, T, 2 * T - 1).permute(1, 0, 2, 3).reshape(B * H, T, 2 * T - 1)
        # rel_shift (attention.py:243-259): column j of row i <- relative position i - j
        Z = B * H
        pad = torch.cat([bd.new_zeros((Z, T, 1)), bd], dim=-1).view(Z, 2 * T, T)
        bd = pad[:, 1:].reshape(Z, T, 2 * T - 1)[:, :, :T]
        drop = self.dropout_rate if self.training else 0.0
        o = ops.AttentionFn.apply(qu, k, v, klen, H, drop, rng.seed(), self._stream, bd.contiguous())
        return self.linear_out(o)


class _BatchNorm1d(RefSchemaModule):
    """nn.BatchNorm1d over the channels of channels-last frames (statistics over batch and time, padded frames included, as
    the reference's unmasked call does)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        if self.training:
            n = x.shape[0] * x.shape[1]
            mean = x.mean((0, 1))
            var = x.var((0, 1), unbiased=False)
            with torch.no_grad():
                self.running_mean.mul_(1 - self.momentum).add_(mean.detach(), alpha=self.momentum)
                self.running_var.mul_(1 - self.momentum).add_(var.detach() * (n / max(n - 1, 1)), alpha=self.momentum)
                self.num_batches_tracked += 1
        else:
            mean, var = self.running_mean, self.running_var
        return (x - mean) * torch.rsqrt(var + self.eps) * self.weight + self.bias


class _DepthwiseConv(RefSchemaModule):
    _ref_layout = {"weight": ("weight", _dw_to_native, _dw_to_ref)}

    def __init__(self, c, k):
        super().__init__()
        self.weight = nn.Parameter(_dw_to_native(_xavier(c, 1, k)))
        self.bias = nn.Parameter(torch.zeros(c))

    def forward(self, x):
        return ops.depthwise_conv(x, self.weight, self.bias)


class ConvolutionModule(nn.Module):
    """_conformer/convolution.py:13-78: pointwise (C -> 2C) -> GLU -> depthwise k -> BatchNorm -> swish -> pointwise."""

    def __init__(self, channels, kernel_size):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        self.pointwise_conv1 = _Conv1dK1(channels, 2 * channels)
        self.depthwise_conv = _DepthwiseConv(channels, kernel_size)
        self.norm = _BatchNorm1d(channels)
        self.pointwise_conv2 = _Conv1dK1(channels, channels)

    def forward(self, x):
        C = x.shape[-1]
        u = self.pointwise_conv1(x)
        u = u[..., :C] * torch.sigmoid(u[..., C:])                                               # GLU over the channel pairs
        u = self.norm(self.depthwise_conv(u.contiguous()))
        return self.pointwise_conv2(u * torch.sigmoid(u))                                        # swish


class EncoderLayer(nn.Module):
    """_conformer/encoder_layer.py:60-174 (normalize_before, macaron, conv module, no concat, no stochastic depth)."""

    def __init__(self, size, self_attn, feed_forward, feed_forward_macaron, conv_module, dropout_rate):
        super().__init__()
        self.self_attn, self.feed_forward, self.feed_forward_macaron, self.conv_module = self_attn, feed_forward, feed_forward_macaron, conv_module
        self.norm_ff, self.norm_mha, self.norm_ff_macaron = FinalNorm(size, 1e-12), FinalNorm(size, 1e-12), FinalNorm(size, 1e-12)
        self.norm_conv, self.norm_final = FinalNorm(size, 1e-12), FinalNorm(size, 1e-12)
        self.dropout_rate = dropout_rate
        self._s = [rng.new_stream() for _ in range(4)]

    def _res(self, x, y, i, scale=1.0):
        """x + scale * dropout(y)"""
        if scale != 1.0:
            y = y * scale
        return ops.dropout_add(y, self.dropout_rate, self.training, self._s[i], res=x)

    def forward(self, x, pos_emb, klen):
        x = self._res(x, self.feed_forward_macaron(self.norm_ff_macaron(x)), 0, 0.5)
        x = self._res(x, self.self_attn(self.norm_mha(x), pos_emb, klen), 1)
        x = self._res(x, self.conv_module(self.norm_conv(x)), 2)
        x = self._res(x, self.feed_forward(self.norm_ff(x)), 3, 0.5)
        return self.norm_final(x)


class Encoder(nn.Module):
    """_conformer/encoder.py (input_layer None, rel_pos): x * sqrt(d) -> dropout; pos_emb -> dropout; blocks; after_norm."""

    def __init__(self, attention_dim, attention_heads, linear_units, num_blocks, dropout_rate, positional_dropout_rate,
                 attention_dropout_rate, cnn_module_kernel):
        super().__init__()
        self.d, self.pos_drop = attention_dim, positional_dropout_rate
        self.encoders = nn.ModuleList([
            EncoderLayer(attention_dim, RelPositionMultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate),
                         MultiLayeredConv1d(attention_dim, linear_units, 1, dropout_rate),
                         MultiLayeredConv1d(attention_dim, linear_units, 1, dropout_rate),
                         ConvolutionModule(attention_dim, cnn_module_kernel), dropout_rate)
            for _ in range(num_blocks)])
        self.after_norm = FinalNorm(attention_dim, 1e-12)
        self._s = [rng.new_stream(), rng.new_stream()]

    def forward(self, xs, klen):
        T = xs.shape[1]
        xs = ops.dropout_add(xs * math.sqrt(self.d), self.pos_drop, self.training, self._s[0])
        pos = rel_positions(T, self.d, xs.device)
        if self.training and self.pos_drop > 0.0 and (2 * T - 1) * self.d % 4 == 0:
            pos = ops.dropout_add(pos, self.pos_drop, True, self._s[1])
        for layer in self.encoders:
            xs = layer(xs, pos, klen)
        return self.after_norm(xs)


class Conformer(nn.Module):
    """generator/modules/conformer.py:9-28.  forward(x (B, T, C), padding_mask (B, T) True = pad) -> (B, T, C)."""

    def __init__(self, dim, attention_heads=2, linear_units=1024, num_blocks=4, dropout_rate=0.2, positional_dropout_rate=0.2,
                 attention_dropout_rate=0.2, normalize_before=True, concat_after=False, positionwise_layer_type="conv1d",
                 positionwise_conv_kernel_size=1, macaron_style=True, pos_enc_layer_type="rel_pos",
                 selfattention_layer_type="rel_selfattn", activation_type="swish", use_cnn_module=True, cnn_module_kernel=7,
                 zero_triu=False, init_type="xavier_uniform", **unused):
        super().__init__()
        assert normalize_before and not concat_after and positionwise_layer_type == "conv1d" and positionwise_conv_kernel_size == 1
        assert macaron_style and pos_enc_layer_type == "rel_pos" and selfattention_layer_type == "rel_selfattn"
        assert activation_type == "swish" and use_cnn_module and not zero_triu, \
            "only the configuration of configs/model/generator/*/conformer.yaml is built"
        self.conformer = Encoder(dim, attention_heads, linear_units, num_blocks, dropout_rate, positional_dropout_rate,
                                 attention_dropout_rate, cnn_module_kernel)

    def forward(self, x, padding_mask):
        klen = (~padding_mask).sum(1).to(torch.int64)
        return self.conformer(x, klen)
