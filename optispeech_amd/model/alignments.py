"""Host-side mirror of optispeech/model/generator/alignments.py: AlignmentModule, GaussianUpsampling,
viterbi_decode, average_by_duration, expand_by_duration -- all device-resident (no host round trips)."""
import torch
from torch import nn

from .. import kernels as K
from .. import ops
from .base import RefSchemaModule, conv_to_native, conv_to_ref


class _Conv1d(RefSchemaModule):
    _ref_layout = {"weight": ("weight", conv_to_native, conv_to_ref)}

    def __init__(self, cin, cout, k, padding):
        super().__init__()
        conv = nn.Conv1d(cin, cout, k)                          # torch default init, as the reference
        self.k, self.padding = k, padding
        self.weight = nn.Parameter(conv_to_native(conv.weight.detach()))
        self.bias = nn.Parameter(conv.bias.detach().clone())

    def forward(self, x, act=None):
        return ops.conv_linear(x, self.weight, self.bias, self.weight.shape[0], self.k, self.padding, act)


class AlignmentModule(nn.Module):
    """AlignmentModule (alignments.py:14-123).  text (B,Tt,adim), feats (B,Tf,odim) channels-last."""

    def __init__(self, adim, odim, cache_prior=True):
        super().__init__()
        self.cache_prior = cache_prior
        self._cache = {}
        self.t_conv1 = _Conv1d(adim, adim, 3, 1)
        self.t_conv2 = _Conv1d(adim, adim, 1, 0)
        self.f_conv1 = _Conv1d(odim, adim, 3, 1)
        self.f_conv2 = _Conv1d(adim, adim, 3, 1)
        self.f_conv3 = _Conv1d(adim, adim, 1, 0)

    def forward(self, text, feats, text_lengths, feats_lengths, x_masks=None):
        t = self.t_conv2(self.t_conv1(text, "relu"))                        # :55-58
        f = self.f_conv3(self.f_conv2(self.f_conv1(feats, "relu"), "relu"))  # :60-64
        prior = self._generate_prior(text_lengths, feats_lengths, feats.shape[1], text.shape[1])
        return ops.AlignLogProbFn.apply(f, t, prior, text_lengths, feats_lengths)   # :66-81

    def _generate_prior(self, text_lengths, feats_lengths, T_feats=None, T_text=None):
        """Beta-binomial prior (B,T_feats,T_text), -inf outside each utterance's block (:85-123); evaluated on the
        device from a log-factorial table instead of scipy on the host."""
        return K.betabinom_prior(text_lengths, feats_lengths, T_feats, T_text)


class GaussianUpsampling(nn.Module):
    """GaussianUpsampling (alignments.py:126-174). Masks are implied by the lengths."""

    def __init__(self, delta=0.1):
        super().__init__()
        self.delta = delta

    def forward(self, hs, ds, x_lengths, y_lengths, T_feats):
        return ops.GaussianUpsampleFn.apply(hs, ds.float(), x_lengths, y_lengths, T_feats, self.delta)


def viterbi_decode(log_p_attn, text_lengths, feats_lengths):
    """viterbi_decode (alignments.py:210-239) -> (durations (B,T_text) f32, path int32 (B,T_feats), bin_item (B,))."""
    path, ds, bin_item = K.mas(log_p_attn.detach(), text_lengths, feats_lengths)
    return ds, path, bin_item


def average_by_duration(ds, xs0, xs1, text_lengths, feats_lengths):
    """average_by_duration (alignments.py:262-280) for pitch and energy in one launch."""
    a0, a1, _ = K.duration_stats(ds, xs0.contiguous(), xs1.contiguous(), text_lengths, feats_lengths)
    return a0, a1


def expand_by_duration(x, durations):
    """expand_by_duration (alignments.py:283-297)."""
    lengths = durations.sum(dim=1)
    return K.expand_by_duration(x, durations.contiguous(), int(lengths.max())), lengths
