"""Host-side mirror of the LightSpeech separable-convolution backbones:
``optispeech/model/generator/modules/lightspeech_transformer.py`` (LightSpeechTransformerEncoder :14-47,
LightSpeechTransformerDecoder :50-96) over ``EncSepConvLayer`` / ``ConvSeparable`` (modules/layers.py:455-506), the
configuration of configs/model/generator/{encoder,decoder}/lightspeech_transformer.yaml.

Same class names, constructor arguments and state-dict keys (``layers.N.layer_norm``, ``layers.N.conv{1,2}.depthwise_conv.weight``
(C, 1, K), ``...pointwise_conv.{weight (C, C, 1), bias}``, ``layer_norm``, decoder ``pos_emb.scale``).  Arithmetic on the HIP
kernels: LayerNorm + padding mask (osp_layernorm_fwd), depthwise conv of width 5..25 (osp_dwconv_fwd / _wgrad, csrc/dwconv.hip),
pointwise conv + bias + ReLU on the conv-GEMM family, dropout (+ residual) from the counter-based RNG (osp_dropout_add).
Frames stay channels-last (B, T, C): the reference's transposes to (T, B, C) / (B, C, T) are layout only.
"""
import math

import torch
from torch import nn

from .. import ops, rng
from .base import RefSchemaModule, conv_to_native, conv_to_ref
from .modules import FinalNorm, _ScaledSinusoidal, row_mask


def _dw_to_native(w):       # (C, 1, K) -> (K, C)
    return w[:, 0, :].t().contiguous()


def _dw_to_ref(w):          # (K, C) -> (C, 1, K)
    return w.t().contiguous()[:, None, :]


class EncSepConvLayer(RefSchemaModule):
    """modules/layers.py:480-506: x + drop(relu(sep2(drop(relu(sep1(mask(LN(x))))))))  with sep = pointwise(depthwise(.))."""

    _ref_layout = {
        "ln_weight": ("layer_norm.weight", None, None), "ln_bias": ("layer_norm.bias", None, None),
        "dw1": ("conv1.depthwise_conv.weight", _dw_to_native, _dw_to_ref),
        "pw1": ("conv1.pointwise_conv.weight", conv_to_native, conv_to_ref), "pb1": ("conv1.pointwise_conv.bias", None, None),
        "dw2": ("conv2.depthwise_conv.weight", _dw_to_native, _dw_to_ref),
        "pw2": ("conv2.pointwise_conv.weight", conv_to_native, conv_to_ref), "pb2": ("conv2.pointwise_conv.bias", None, None),
    }

    def __init__(self, c, kernel_size, dropout, activation="relu"):
        super().__init__()
        assert activation == "relu", "configs/model/generator/*/lightspeech_transformer.yaml: activation relu"
        assert kernel_size % 2 == 1
        self.c, self.kernel_size, self.dropout = c, kernel_size, float(dropout)
        std = math.sqrt((4 * (1.0 - dropout)) / (kernel_size * c))          # ConvSeparable init, layers.py:467-470
        self.ln_weight, self.ln_bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
        for i in (1, 2):
            setattr(self, f"dw{i}", nn.Parameter(torch.randn(kernel_size, c) * std))
            setattr(self, f"pw{i}", nn.Parameter(torch.randn(c, 1, c) * std))
            setattr(self, f"pb{i}", nn.Parameter(torch.zeros(c)))
        self._s1, self._s2 = rng.new_stream(), rng.new_stream()

    def forward(self, x, rowmask):
        """x (B, T, C) channels-last; rowmask (B*T,) float keep mask or None (encoder_padding_mask of the reference)."""
        h = ops.layer_norm(x, self.ln_weight, self.ln_bias, 1e-6, rowmask=rowmask)          # LN, then masked_fill(pad, 0)
        h = ops.conv_linear(ops.depthwise_conv(h, self.dw1), self.pw1, self.pb1, self.c, act="relu")
        h = ops.dropout_add(h, self.dropout, self.training, self._s1)
        h = ops.conv_linear(ops.depthwise_conv(h, self.dw2), self.pw2, self.pb2, self.c, act="relu")
        return ops.dropout_add(h, self.dropout, self.training, self._s2, res=x)


class LightSpeechTransformerEncoder(nn.Module):
    """lightspeech_transformer.py:14-47.  forward(x (B, T, C), padding_mask (B, T) True = pad) -> (B, T, C)."""

    def __init__(self, dim, kernel_sizes, activation="relu", dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList([EncSepConvLayer(dim, k, dropout, activation) for k in kernel_sizes])
        self.layer_norm = FinalNorm(dim, 1e-12)                                  # layers.LayerNorm: eps 1e-12 (layers.py:34)

    def forward(self, x, padding_mask):
        rm = row_mask(padding_mask)
        for layer in self.layers:
            x = layer(x, rm)
        return ops.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, 1e-12, rowmask=rm)      # LN, then * (1 - pad)


class LightSpeechTransformerDecoder(nn.Module):
    """lightspeech_transformer.py:50-96 (require_w = False): (x + pos) * mask -> dropout -> layers -> nn.LayerNorm."""

    def __init__(self, dim, kernel_sizes, activation="relu", dropout=0.2, max_source_positions=2000):
        super().__init__()
        self.pos_emb = _ScaledSinusoidal(dim, theta=max_source_positions)
        self.layers = nn.ModuleList([EncSepConvLayer(dim, k, dropout, activation) for k in kernel_sizes])
        self.dropout = float(dropout)
        self.layer_norm = FinalNorm(dim, 1e-5)                                   # nn.LayerNorm default eps
        self._stream = rng.new_stream()

    def forward(self, x, padding_mask, *, require_w=False):
        assert not require_w, "the separable-conv layers have no attention weights to return"
        B, T, C = x.shape
        rm = row_mask(padding_mask)
        pos = self.pos_emb.table_for(T, x.device)[:T] * self.pos_emb.scale       # (T, C) * learnable scale
        x = (x + pos[None]) * rm.view(B, T, 1)
        x = ops.dropout_add(x, self.dropout, self.training, self._stream)
        for layer in self.layers:
            x = layer(x, rm)
        return ops.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, 1e-5)
