"""Host-side mirror of the Transformer backbone variant (BASELINE config 4):
``optispeech/model/generator/modules/transformer.py`` wrapping the ESPnet-style encoder
(``_transformer/encoder.py``, ``encoder_layer.py``, ``attention.py``, ``multi_layer_conv.py``, ``embedding.py``).

Same class names and state-dict keys (``transformer.embed.0.alpha``, ``transformer.encoders.N.{self_attn.linear_*,
feed_forward.w_*, norm1, norm2}``, ``transformer.after_norm``); pre-LN, 2 heads, conv1d-k1 feed-forward, scaled positional
encoding -- the configuration of configs/model/generator/{encoder,decoder}/transformer.yaml.  Arithmetic: LayerNorm /
linear layers / attention GEMMs and the masked softmax run in the HIP kernels (ops.layer_norm, ops.conv_linear,
ops.AttentionFn); dropout (+ residual) on osp_dropout_add, the scaled positional encoding on osp_posenc_fwd.
"""
import math

import torch
from torch import nn

from .. import ops, rng
from .base import RefSchemaModule, conv_to_native, conv_to_ref
from .modules import FinalNorm


def _xavier(*shape):
    w = torch.empty(*shape)
    nn.init.xavier_uniform_(w)                                   # _transformer/initialize.py (init_type xavier_uniform)
    return w


class _Linear(RefSchemaModule):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(_xavier(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x, act=None):
        return ops.conv_linear(x, self.weight, self.bias, self.weight.shape[0], 1, 0, act)


class _Conv1dK1(RefSchemaModule):
    """torch.nn.Conv1d(cin, cout, 1): weight (cout, cin, 1) in the reference, (cout, 1, cin) natively."""
    _ref_layout = {"weight": ("weight", conv_to_native, conv_to_ref)}

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(conv_to_native(_xavier(cout, cin, 1)))
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x, act=None):
        return ops.conv_linear(x, self.weight, self.bias, self.weight.shape[0], 1, 0, act)


def _dropout(x, p, training, stream_id, res=None):
    """res + F.dropout(x): one launch of the counter-based dropout kernel (osp_dropout_add), the same kernel on the gradient in the
    backward.  (Through round 5 every site materialised a keep mask -- three fills, a LayerNorm launch and an ATen multiply per site,
    26 sites per forward -- with the same Philox counters: the values are bit-identical.)"""
    return ops.dropout_add(x, p, training, stream_id, res=res)


class ScaledPositionalEncoding(nn.Module):
    """_transformer/embedding.py:91-124: x + alpha * pe, then dropout.  The table is a non-persistent buffer (it moves with the module
    and never enters a state dict, like the reference's ``self.pe``); on the GPU the sum is one launch that reads alpha on the
    device (ops.ScaledPosEncFn), so a call tape can hold the whole forward."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__()
        self.d_model, self.dropout_rate = d_model, dropout_rate
        self.alpha = nn.Parameter(torch.tensor(1.0))
        self.register_buffer("_pe", self._table(max_len), persistent=False)
        self._stream = rng.new_stream()

    def _table(self, rows):
        pos = torch.arange(0, rows, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / self.d_model))
        pe = torch.zeros(rows, self.d_model)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        return pe

    def pe(self, T, device):
        if self._pe.shape[0] < T or self._pe.device != device:        # extend_pe (embedding.py:58-77)
            self._pe = self._table(max(T, self._pe.shape[0])).to(device)
        return self._pe[:T]

    def forward(self, x):
        pe = self.pe(x.shape[1], x.device)
        if x.is_cuda and x.dtype == torch.float32:
            y = ops.ScaledPosEncFn.apply(x, self.alpha, pe)
        else:
            y = x + self.alpha * pe
        return _dropout(y, self.dropout_rate, self.training, self._stream)


class MultiHeadedAttention(nn.Module):
    """_transformer/attention.py:13-125 (self-attention with a key-padding mask)."""

    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k, self.h, self.dropout_rate = n_feat // n_head, n_head, dropout_rate
        self.linear_q, self.linear_k = _Linear(n_feat, n_feat), _Linear(n_feat, n_feat)
        self.linear_v, self.linear_out = _Linear(n_feat, n_feat), _Linear(n_feat, n_feat)
        self._stream = rng.new_stream()

    def forward(self, x, klen):
        q, k, v = self.linear_q(x), self.linear_k(x), self.linear_v(x)
        p = self.dropout_rate if self.training else 0.0
        o = ops.AttentionFn.apply(q, k, v, klen, self.h, p, rng.seed(), self._stream)
        return self.linear_out(o)


class MultiLayeredConv1d(nn.Module):
    """_transformer/multi_layer_conv.py:11-62 with kernel_size 1: w_2(dropout(relu(w_1 x)))."""

    def __init__(self, in_chans, hidden_chans, kernel_size, dropout_rate):
        super().__init__()
        assert kernel_size == 1, "configs/model/generator/*/transformer.yaml: positionwise_conv_kernel_size 1"
        self.w_1, self.w_2, self.dropout_rate = _Conv1dK1(in_chans, hidden_chans), _Conv1dK1(hidden_chans, in_chans), dropout_rate
        self._stream = rng.new_stream()

    def forward(self, x):
        return self.w_2(_dropout(self.w_1(x, act="relu"), self.dropout_rate, self.training, self._stream))


class EncoderLayer(nn.Module):
    """_transformer/encoder_layer.py:60-116, normalize_before = True, concat_after = False, no stochastic depth."""

    def __init__(self, size, self_attn, feed_forward, dropout_rate):
        super().__init__()
        self.self_attn, self.feed_forward = self_attn, feed_forward
        self.norm1, self.norm2 = FinalNorm(size, 1e-12), FinalNorm(size, 1e-12)      # _transformer/layer_norm.py:20-23
        self.dropout_rate = dropout_rate
        self._s1, self._s2 = rng.new_stream(), rng.new_stream()

    def forward(self, x, klen):
        x = _dropout(self.self_attn(self.norm1(x), klen), self.dropout_rate, self.training, self._s1, res=x)
        return _dropout(self.feed_forward(self.norm2(x)), self.dropout_rate, self.training, self._s2, res=x)


class Encoder(nn.Module):
    """_transformer/encoder.py (input_layer None, pos_enc_class ScaledPositionalEncoding): embed -> encoders -> after_norm."""

    def __init__(self, attention_dim, attention_heads, linear_units, num_blocks, dropout_rate, positional_dropout_rate,
                 attention_dropout_rate):
        super().__init__()
        self.embed = nn.Sequential(ScaledPositionalEncoding(attention_dim, positional_dropout_rate))
        self.encoders = nn.ModuleList([
            EncoderLayer(attention_dim, MultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate),
                         MultiLayeredConv1d(attention_dim, linear_units, 1, dropout_rate), dropout_rate)
            for _ in range(num_blocks)])
        self.after_norm = FinalNorm(attention_dim, 1e-12)

    def forward(self, xs, klen):
        xs = self.embed(xs)
        for layer in self.encoders:
            xs = layer(xs, klen)
        return self.after_norm(xs)


class Transformer(nn.Module):
    """generator/modules/transformer.py:9-27.  forward(x (B, T, C), padding_mask (B, T) True = pad) -> (B, T, C)."""

    def __init__(self, dim, attention_heads=2, linear_units=1024, num_blocks=4, dropout_rate=0.2, positional_dropout_rate=0.2,
                 attention_dropout_rate=0.2, normalize_before=True, concat_after=False, positionwise_layer_type="conv1d",
                 positionwise_conv_kernel_size=1, use_scaled_pos_enc=True, init_alpha=1.0, init_type="xavier_uniform", **unused):
        super().__init__()
        assert normalize_before and not concat_after and positionwise_layer_type == "conv1d" and use_scaled_pos_enc, \
            "only the configuration of configs/model/generator/*/transformer.yaml is built"
        assert positionwise_conv_kernel_size == 1
        self.transformer = Encoder(dim, attention_heads, linear_units, num_blocks, dropout_rate, positional_dropout_rate,
                                   attention_dropout_rate)
        with torch.no_grad():
            self.transformer.embed[-1].alpha.fill_(init_alpha)

    def forward(self, x, padding_mask):
        # mask = ~padding_mask (prefix-valid, as sequence_mask builds it): the valid-key counts are the lengths the mask was built
        # from when the generator built it (generator.padding_mask leaves them on the tensor), a row sum otherwise
        tag = getattr(padding_mask, "_osp_lengths", None)
        if tag is not None and tag[0] == (-1 if padding_mask.is_inference() else padding_mask._version) and tag[1].is_cuda == x.is_cuda:
            # (raw lengths may exceed T -- an inconsistent batch, a mask built for a longer T and sliced -- where a row sum could
            # not: every attention kernel clamps, kl = min(T, klen[b]) (csrc/attention.hip:20,134, attention_train.hip:190,303,407;
            # tests/test_gpu_attention*.py::..._lengths_beyond_T), so no launch is spent on a clamp here: an ATen op would also
            # poison the taped acoustic-model region)
            klen = tag[1]
        else:
            klen = (~padding_mask).sum(1).to(torch.int64)
        return self.transformer(x, klen)
