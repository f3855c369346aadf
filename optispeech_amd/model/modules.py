"""Host-side mirror of optispeech/model/generator/modules/{convnext,core,layers}.py for the ConvNeXt
configuration: same constructor contracts and state-dict keys, every forward/backward a HIP kernel
sequence.  Activations are channels-last (B, T, C) end-to-end (the reference flips layouts per block).
"""
import math
from typing import Optional

import torch
from torch import nn

from .. import kernels as K
from .. import ops, precision, rng
from .base import RefSchemaModule, conv_to_native, conv_to_ref, dw_to_native, dw_to_ref


def row_mask(padding_mask):
    """(B,T) bool True=pad  ->  (B*T,) float keep mask (1 - padding_mask.float(), convnext.py:95)."""
    if padding_mask is None:
        return None
    # one mask tensor is handed to several modules of a forward (encoder, three predictors, two embeds ...): convert it once and
    # keep the result on the tensor object (it dies with the mask; a bool mask is never written in place on the path)
    # inference tensors keep no version counter; a conversion done eagerly (a graph's warm-up) must not satisfy a capture, or the
    # replayed graph would keep reading the warm-up's mask
    ver = (-1 if padding_mask.is_inference() else padding_mask._version,
           padding_mask.is_cuda and torch.cuda.is_current_stream_capturing())
    hit = getattr(padding_mask, "_osp_rowmask", None)
    if hit is None or hit[0] != ver:
        hit = (ver, (~padding_mask).to(torch.float32).reshape(-1).contiguous())
        try:
            padding_mask._osp_rowmask = hit
        except (AttributeError, RuntimeError):
            pass
    return hit[1]


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

    def forward(self, x, rowmask=None, rowscale=None, rowf=None):
        """x (B,T,C) channels-last; rowmask (B*T,) keep mask applied after the residual (convnext.py:99-101); rowscale (B*T,):
        this block's DropPath factors when the backbone drew them for all blocks at once (None = draw here)."""
        if rowscale is None and self.training and self.drop_prob > 0.0:         # DropPath, convnext.py:121-129
            B, T, _ = x.shape
            keep = 1.0 - self.drop_prob
            r = torch.empty((B, 1), device=x.device, dtype=torch.float32).bernoulli_(keep) / keep
            rowscale = r.expand(B, T).reshape(-1).contiguous()
        return ops.ConvNeXtBlockFn.apply(x, self.dwconv_weight, self.dwconv_bias, self.norm_weight, self.norm_bias,
                                         self.pwconv1_weight, self.pwconv1_bias, self.pwconv2_weight,
                                         self.pwconv2_bias, self.gamma, rowmask, rowscale, rowf)


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
        self._drop_stream = rng.new_stream()                 # Philox stream of this backbone's DropPath draws

    def forward(self, x, padding_mask=None):
        rm = row_mask(padding_mask)
        if precision.is_bf16() and x.is_cuda:
            # every bf16 weight copy the blocks of this backbone will ask for in this optimiser epoch (forward: W1, W2; backward:
            # W1^T and gamma * W2^T), packed by ONE launch instead of 3-4 per block
            need_bwd = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
            req = []
            for blk in self.convnext:
                if blk.intermediate_dim % 64 == 0 and blk.dim % 64 == 0:
                    req += [(blk.pwconv1_weight, False, None), (blk.pwconv2_weight, False, None)]
                    if need_bwd:
                        req += [(blk.pwconv1_weight, True, None), (blk.pwconv2_weight, True, blk.gamma)]
            if req:
                K.param_bf16_many(req)
        scales, rowfs = self._drop_path_scales(x, rm) if self.training else (None, None)
        for i, blk in enumerate(self.convnext):
            x = blk(x, rm, None if scales is None else scales[i], None if rowfs is None else rowfs[i])
        return self.final_layer_norm(x)

    def _drop_path_scales(self, x, rm):
        """DropPath factors of ALL blocks (convnext.py:121-129 draws bernoulli(keep) / keep per block and utterance) and their
        product with the padding mask, from the package's counter-based RNG in ONE launch: (scale (L, B*T), rowf (L, B*T) or None);
        (None, None) when no block drops."""
        drops = [b.drop_prob for b in self.convnext]
        if not any(p > 0.0 for p in drops):
            return None, None
        B, T, _ = x.shape
        return K.drop_path_rows(drops, rm, B, T, rng.seed(), self._drop_stream, x.device)


# =================================================================================================== text embedding
def _sinusoid_table(T, dim, theta):
    """ScaledSinusoidalEmbedding buffers (modules/layers.py:54-70) as a (T, dim) constant table, built on the
    host exactly as the reference does (inv_freq = theta ** -(arange(half)/half); cat(sin, cos))."""
    half = dim // 2
    inv_freq = theta ** -(torch.arange(half).float() / half)
    ang = torch.arange(T).float()[:, None] * inv_freq[None, :]
    return torch.cat((ang.sin(), ang.cos()), dim=-1).contiguous()


class _ScaledSinusoidal(RefSchemaModule):
    def __init__(self, dim, theta):
        super().__init__()
        assert dim % 2 == 0
        self.dim, self.theta = dim, theta
        self.scale = nn.Parameter(torch.ones(1) * dim ** -0.5)
        self.register_buffer("table", _sinusoid_table(256, dim, theta), persistent=False)

    def table_for(self, T, device):
        if self.table.shape[0] < T or self.table.device != device:
            self.table = _sinusoid_table(max(T, 2 * self.table.shape[0]), self.dim, self.theta).to(device)
        return self.table


class _EmbedTokens(RefSchemaModule):
    def __init__(self, n_vocab, dim, padding_idx):
        super().__init__()
        self.padding_idx = padding_idx
        self.weight = nn.Parameter(torch.empty(n_vocab, dim))
        nn.init.normal_(self.weight)                        # nn.Embedding default init
        with torch.no_grad():
            self.weight[padding_idx].zero_()


class TextEmbedding(nn.Module):
    """TextEmbedding (modules/core.py:11-31). forward(src_tokens (B,T) int64) -> (x (B,T,dim), None)."""

    def __init__(self, dim: int, n_vocab: int, dropout: float = 0.0, padding_idx: int = 0,
                 max_source_positions: int = 2000):
        super().__init__()
        assert padding_idx == 0
        self.dim, self.dropout = dim, float(dropout)
        self.embed_tokens = _EmbedTokens(n_vocab, dim, padding_idx)
        self.embed_positions = _ScaledSinusoidal(dim, theta=max_source_positions)
        self._stream = rng.new_stream()

    def forward(self, src_tokens):
        p = self.dropout if self.training else 0.0
        pos = self.embed_positions.table_for(src_tokens.shape[1], src_tokens.device)
        x = ops.TextEmbedFn.apply(src_tokens.contiguous(), self.embed_tokens.weight, self.embed_positions.scale, pos, p,
                                  rng.seed(), self._stream)
        return x, None


# =================================================================================================== variance predictors
class _ConvLayer(RefSchemaModule):
    """`conv.{i}` Sequential of the reference: .0 = Conv1d, .2 = LayerNorm(dim=1) (core.py:62-76)."""
    _ref_layout = {"conv_weight": ("0.weight", conv_to_native, conv_to_ref), "conv_bias": ("0.bias", None, None),
                   "ln_weight": ("2.weight", None, None), "ln_bias": ("2.bias", None, None)}

    def __init__(self, cin, cout, k, dropout):
        super().__init__()
        self.k, self.dropout = k, float(dropout)
        conv = nn.Conv1d(cin, cout, k)                      # reference default init (kaiming-uniform)
        self.conv_weight = nn.Parameter(conv_to_native(conv.weight.detach()))
        self.conv_bias = nn.Parameter(conv.bias.detach().clone())
        self.ln_weight = nn.Parameter(torch.ones(cout))
        self.ln_bias = nn.Parameter(torch.zeros(cout))
        self._stream = rng.new_stream()

    def forward(self, x):
        p = self.dropout if self.training else 0.0
        return ops.PredictorLayerFn.apply(x, self.conv_weight, self.conv_bias, self.ln_weight, self.ln_bias, self.k, p,
                                          rng.seed(), self._stream)


class _Linear(RefSchemaModule):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        lin = nn.Linear(cin, cout, bias)
        self.weight = nn.Parameter(lin.weight.detach().clone())
        self.bias = nn.Parameter(lin.bias.detach().clone()) if bias else None


class VariancePredictor(nn.Module):
    """VariancePredictor (modules/core.py:34-97). forward(x (B,T,dim), padding_mask (B,T) True=pad) -> (B,T)."""

    def __init__(self, dim: int, num_layers: int, intermediate_dim: int, kernel_size: int, dropout: float = 0.1,
                 conv_layer_class: type = torch.nn.Conv1d):
        super().__init__()
        if conv_layer_class is not torch.nn.Conv1d:
            raise ValueError("only torch.nn.Conv1d predictors are on the ConvNeXt hot path")
        self.dim, self.conv_layer_class = dim, conv_layer_class
        self.conv = nn.ModuleList([_ConvLayer(dim if i == 0 else intermediate_dim, intermediate_dim, kernel_size, dropout)
                                   for i in range(num_layers)])
        self.linear = _Linear(intermediate_dim, 1)

    def forward(self, x, padding_mask):
        for layer in self.conv:
            x = layer(x)
        y = ops.conv_linear(x, self.linear.weight, self.linear.bias, 1, rowmask=row_mask(padding_mask))
        return y.squeeze(-1)


class DurationPredictor(VariancePredictor):
    """DurationPredictor (modules/core.py:100-133)."""

    def __init__(self, *args, clip_val=1e-8, **kwargs):
        super().__init__(*args, **kwargs)
        self.clip_val = clip_val

    @torch.inference_mode()
    def infer(self, x, mask, factor=1.0):
        log_d = self(x, mask)
        d = torch.exp(log_d) - self.clip_val
        d = torch.ceil(d * factor)
        d = torch.clamp(d.long(), min=0)
        return d.masked_fill(mask, 0)


class _EmbedConv(RefSchemaModule):
    """`embed` Sequential of the reference: .0 = Conv1d(1, dim, k) (+ Dropout) (core.py:141-149)."""
    _ref_layout = {"weight": ("0.weight", conv_to_native, conv_to_ref), "bias": ("0.bias", None, None)}

    def __init__(self, dim, k):
        super().__init__()
        conv = nn.Conv1d(1, dim, k)
        self.weight = nn.Parameter(conv_to_native(conv.weight.detach()))
        self.bias = nn.Parameter(conv.bias.detach().clone())


class PitchPredictor(nn.Module):
    """PitchPredictor (modules/core.py:136-176)."""

    def __init__(self, *args, embed_kernel_size=9, embed_dropout=0.1, **kwargs):
        super().__init__()
        self.predictor = VariancePredictor(*args, **kwargs)
        self.dim = kwargs["dim"]
        self.conv_layer_class = kwargs.get("conv_layer_class", torch.nn.Conv1d)
        self.embed = _EmbedConv(self.dim, embed_kernel_size)
        self.embed_dropout = float(embed_dropout)
        self._stream = rng.new_stream()

    def _add(self, x, values, padding_mask):
        p = self.embed_dropout if self.training else 0.0
        return ops.VarianceEmbedFn.apply(x, values, self.embed.weight, self.embed.bias, row_mask(padding_mask), p,
                                         rng.seed(), self._stream)

    def forward(self, x, padding_mask, target):
        preds = self.predictor(x, padding_mask)
        return self._add(x, target, padding_mask), preds

    @torch.inference_mode()
    def infer(self, x, padding_mask, factor=1.0):
        preds = self.predictor(x, padding_mask) * factor
        return self._add(x, preds, padding_mask), preds


class EnergyPredictor(PitchPredictor):
    pass
