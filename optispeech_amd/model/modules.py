"""Host-side mirror of optispeech/model/generator/modules/{convnext,core,layers}.py for the ConvNeXt
configuration: same constructor contracts and state-dict keys, every forward/backward a HIP kernel
sequence.  Activations are channels-last (B, T, C) end-to-end (the reference flips layouts per block).
"""
import math
from typing import Optional

import torch
from torch import nn

from .. import ops
from .base import RefSchemaModule, conv_to_native, conv_to_ref, dw_to_native, dw_to_ref


def row_mask(padding_mask):
    """(B,T) bool True=pad  ->  (B*T,) float keep mask (1 - padding_mask.float(), convnext.py:95)."""
    if padding_mask is None:
        return None
    return (~padding_mask).to(torch.float32).reshape(-1).contiguous()


class ConvNeXtBlock(RefSchemaModule):
    """ConvNeXtBlock (generator/modules/convnext.py:8-47): constructor args and keys as the reference."""

    _ref_layout = {
        "dwconv_weight": ("dwconv.weight", dw_to_native, dw_to_ref), "dwconv_bias": ("dwconv.bias", None, None),
        "norm_weight": ("norm.weight", None, None), "norm_bias": ("norm.bias", None, None),
        "pwconv1_weight": ("pwconv1.weight", None, None), "pwconv1_bias": ("pwconv1.bias", None, None),
        "pwconv2_weight": ("pwconv2.weight", None, None), "pwconv2_bias": ("pwconv2.bias", None, None),
    }

    def __init__(self, dim: int, intermediate_dim: int, drop_path: float = 0.0, layer_scale_init_value: float = None):
        super().__init__()
        self.dim, self.intermediate_dim, self.drop_prob = dim, intermediate_dim, float(drop_path)
        self.dwconv_weight = nn.Parameter(torch.empty(7, dim))
        self.dwconv_bias = nn.Parameter(torch.zeros(dim))
        self.norm_weight = nn.Parameter(torch.ones(dim))
        self.norm_bias = nn.Parameter(torch.zeros(dim))
        self.pwconv1_weight = nn.Parameter(torch.empty(intermediate_dim, dim))
        self.pwconv1_bias = nn.Parameter(torch.zeros(intermediate_dim))
        self.pwconv2_weight = nn.Parameter(torch.empty(dim, intermediate_dim))
        self.pwconv2_bias = nn.Parameter(torch.zeros(dim))
        if not (layer_scale_init_value and layer_scale_init_value > 0):
            raise ValueError("layer_scale_init_value must be > 0 (the reference compares it with 0, convnext.py:29)")
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim))
        for w in (self.dwconv_weight, self.pwconv1_weight, self.pwconv2_weight):
            nn.init.trunc_normal_(w, std=0.02)                                   # convnext.py:87-90

    def forward(self, x, rowmask=None):
        """x (B,T,C) channels-last; rowmask (B*T,) keep mask applied after the residual (convnext.py:99-101)."""
        rowscale = None
        if self.training and self.drop_prob > 0.0:                              # DropPath, convnext.py:121-129
            B, T, _ = x.shape
            keep = 1.0 - self.drop_prob
            r = torch.empty((B, 1), device=x.device, dtype=torch.float32).bernoulli_(keep) / keep
            rowscale = r.expand(B, T).reshape(-1).contiguous()
        return ops.ConvNeXtBlockFn.apply(x, self.dwconv_weight, self.dwconv_bias, self.norm_weight, self.norm_bias,
                                         self.pwconv1_weight, self.pwconv1_bias, self.pwconv2_weight,
                                         self.pwconv2_bias, self.gamma, rowmask, rowscale)


class FinalNorm(RefSchemaModule):
    """nn.LayerNorm(dim, eps) with the reference's `weight`/`bias` keys."""

    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x, rowmask=None):
        return ops.layer_norm(x, self.weight, self.bias, self.eps, rowmask)


class ConvNeXtBackbone(nn.Module):
    """ConvNeXtBackbone (generator/modules/convnext.py:50-103). forward(x (B,T,C), padding_mask (B,T) True=pad)."""

    def __init__(self, dim: int, intermediate_dim: int, num_layers: int, drop_path: float = 0.0,
                 layer_scale_init_value: Optional[float] = None):
        super().__init__()
        layer_scale_init_value = layer_scale_init_value or 1 / num_layers
        rates = [v.item() for v in torch.linspace(0, drop_path, num_layers)]      # convnext.py:72
        self.convnext = nn.ModuleList([ConvNeXtBlock(dim, intermediate_dim, r, layer_scale_init_value) for r in rates])
        self.final_layer_norm = FinalNorm(dim, 1e-6)

    def forward(self, x, padding_mask=None):
        rm = row_mask(padding_mask)
        for blk in self.convnext:
            x = blk(x, rm)
        return self.final_layer_norm(x)
