"""Host-side mirror of optispeech/model/vocoder/wavenext/__init__.py (WaveNeXt + WaveNeXtHead)."""
from typing import Optional

import torch
from torch import nn

from .. import ops
from .base import RefSchemaModule, conv_to_native, conv_to_ref
from .modules import ConvNeXtBackbone, FinalNorm, row_mask

# the head's (n_fft + 2 = 1026)-wide hidden is stored padded with zeros to a multiple of 64 (1088): as the reduction dimension of
# linear_2 and of linear_1's input gradient it then is a whole number of 64-deep MFMA k-slabs with 16-byte rows (the LDS-DMA
# kernels); a 1028-wide hidden took the element-wise generic loader (145 us per GEMM instead of ~15).  The pad weights / biases
# are zero, receive exactly zero gradients (their inputs / upstream gradients are zero) and never enter a state dict.
_PAD = 64


def _pad_rows(w, n):
    return torch.cat([w, w.new_zeros((n - w.shape[0],) + tuple(w.shape[1:]))], 0) if w.shape[0] < n else w


class _Embed(RefSchemaModule):
    _ref_layout = {"weight": ("weight", conv_to_native, conv_to_ref)}

    def __init__(self, cin, cout, k):
        super().__init__()
        conv = nn.Conv1d(cin, cout, k)
        self.k = k
        self.weight = nn.Parameter(conv_to_native(conv.weight.detach()))
        self.bias = nn.Parameter(conv.bias.detach().clone())


class _Linear1(RefSchemaModule):
    """linear_1 (dim -> n_fft + 2) stored with zero rows up to a multiple of _PAD outputs."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cout = cout
        self.cout_p = (cout + _PAD - 1) // _PAD * _PAD
        w = torch.zeros(self.cout_p, cin)
        nn.init.trunc_normal_(w[:cout], std=0.02)
        lin = nn.Linear(cin, cout)
        b = torch.zeros(self.cout_p)
        b[:cout] = lin.bias.detach()
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(b)
        self._ref_layout = {"weight": ("weight", lambda v: _pad_rows(v, self.cout_p), lambda v: v[: self.cout]),
                            "bias": ("bias", lambda v: _pad_rows(v, self.cout_p), lambda v: v[: self.cout])}


class _Linear2(RefSchemaModule):
    """linear_2 (n_fft + 2 -> hop, no bias) stored with zero columns up to a multiple of _PAD inputs."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin = cin
        self.cin_p = (cin + _PAD - 1) // _PAD * _PAD
        w = torch.zeros(cout, self.cin_p)
        nn.init.trunc_normal_(w[:, :cin], std=0.02)
        self.weight = nn.Parameter(w)
        self._ref_layout = {"weight": ("weight", lambda v: _pad_rows(v.t(), self.cin_p).t().contiguous(),
                                       lambda v: v[:, : self.cin])}


class WaveNeXtHead(nn.Module):
    """WaveNeXtHead (wavenext/__init__.py:9-48). forward(x (B,L,H)) -> (B, L*hop) clipped to [-1,1]."""

    def __init__(self, dim: int, n_fft: int, hop_length: int):
        super().__init__()
        self.linear_1 = _Linear1(dim, n_fft + 2)
        self.linear_2 = _Linear2(n_fft + 2, hop_length)

    def forward(self, x):
        B = x.shape[0]
        h = ops.conv_linear(x, self.linear_1.weight, self.linear_1.bias, self.linear_1.cout_p, out_bf16=True)
        a = ops.conv_linear(h, self.linear_2.weight, None, self.linear_2.weight.shape[0])
        a = a.reshape(B, -1)
        if a.is_cuda and a.dtype == torch.float32:
            return ops.ClipFn.apply(a, -1.0, 1.0)                       # :47 on the HIP kernel (was torch glue)
        return torch.clip(a, min=-1.0, max=1.0)


class WaveNeXt(nn.Module):
    """WaveNeXt (wavenext/__init__.py:51-86). forward(x (B,T,Cin) channels-last, f0 ignored, padding_mask (B,T))."""

    def __init__(self, input_channels: int, dim: int, intermediate_dim: int, num_layers: int, n_fft: int,
                 hop_length: int, sample_rate: int, drop_path: float = 0.0,
                 layer_scale_init_value: Optional[float] = None):
        super().__init__()
        self.embed = _Embed(input_channels, dim, 7)
        self.norm = FinalNorm(dim, 1e-6)
        self.backbone = ConvNeXtBackbone(dim=dim, intermediate_dim=intermediate_dim, num_layers=num_layers,
                                         drop_path=drop_path, layer_scale_init_value=layer_scale_init_value)
        self.head = WaveNeXtHead(dim=dim, n_fft=n_fft, hop_length=hop_length)

    def forward(self, x, f0=None, padding_mask=None):
        h = ops.conv_linear(x, self.embed.weight, self.embed.bias, self.embed.weight.shape[0], 7, 3)
        h = self.norm(h)
        h = self.backbone(h, padding_mask)
        return self.head(h)
