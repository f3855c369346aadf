"""Host-side mirror of the LeanSpeech backbone: ``optispeech/model/generator/modules/leanspeech.py``
(LeanSpeechBackbone :14-39, LeanSpeechBlock :42-63, ConvGLU :66-97), configuration of
configs/model/generator/{encoder,decoder}/leanspeech.yaml (kernel_size 9, 4 layers, drop_path 0.2).

Same class names, constructor arguments and state-dict keys (``layers.N.lstm.{weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0}``,
``layers.N.conv.conv.0.{depthwise_conv.weight, pointwise_conv.weight, pointwise_conv.bias}``, ``layers.N.conv.conv.1.{weight, bias}``,
``layers.N.final_layer_norm``).  Arithmetic on the HIP kernels: the LSTM (input projection / gradients on the conv-GEMM family,
the recurrence in csrc/lstm.hip), the separable conv (csrc/dwconv.hip + pointwise GEMM), both LayerNorms; tanh / GLU / the
residual sums are element-wise torch glue.
"""
import math

import torch
from torch import nn

from .. import ops, rng
from .base import RefSchemaModule, conv_to_native, conv_to_ref
from .lightspeech import _dw_to_native, _dw_to_ref
from .modules import row_mask


class LeanSpeechBlock(RefSchemaModule):
    """leanspeech.py:42-63:  x + drop_path(LN(tanh(lstm(x)) + (x + GLU(LN_2C(sepconv(x)))) * mask))."""

    _ref_layout = {
        "w_ih": ("lstm.weight_ih_l0", None, None), "w_hh": ("lstm.weight_hh_l0", None, None),
        "b_ih": ("lstm.bias_ih_l0", None, None), "b_hh": ("lstm.bias_hh_l0", None, None),
        "dw": ("conv.conv.0.depthwise_conv.weight", _dw_to_native, _dw_to_ref),
        "pw": ("conv.conv.0.pointwise_conv.weight", conv_to_native, conv_to_ref), "pb": ("conv.conv.0.pointwise_conv.bias", None, None),
        "gln_weight": ("conv.conv.1.weight", None, None), "gln_bias": ("conv.conv.1.bias", None, None),
        "ln_weight": ("final_layer_norm.weight", None, None), "ln_bias": ("final_layer_norm.bias", None, None),
    }

    def __init__(self, dim, kernel_size, drop_path=0.0):
        super().__init__()
        assert kernel_size % 2 == 1 and dim % 32 == 0
        self.dim, self.kernel_size, self.drop_prob = dim, kernel_size, float(drop_path)
        lstm = nn.LSTM(dim, dim, num_layers=1, batch_first=True)                 # torch's default init
        self.w_ih, self.w_hh = nn.Parameter(lstm.weight_ih_l0.detach().clone()), nn.Parameter(lstm.weight_hh_l0.detach().clone())
        self.b_ih, self.b_hh = nn.Parameter(lstm.bias_ih_l0.detach().clone()), nn.Parameter(lstm.bias_hh_l0.detach().clone())
        std = math.sqrt(4.0 / (kernel_size * 2 * dim))                           # ConvSeparable init (layers.py:467-470), dropout 0
        self.dw = nn.Parameter(torch.randn(kernel_size, dim) * std)
        self.pw = nn.Parameter(torch.randn(2 * dim, 1, dim) * std)
        self.pb = nn.Parameter(torch.zeros(2 * dim))
        self.gln_weight, self.gln_bias = nn.Parameter(torch.ones(2 * dim)), nn.Parameter(torch.zeros(2 * dim))
        self.ln_weight, self.ln_bias = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.zeros(dim))
        self._drop_stream = rng.new_stream()

    def forward(self, x, rowmask):
        B, T, C = x.shape
        lx = torch.tanh(ops.lstm(x, self.w_ih, self.w_hh, self.b_ih, self.b_hh))
        # ConvGLU: inputs + GLU(LayerNorm over the 2C channels (eps 1e-12) of the separable conv)
        u = ops.conv_linear(ops.depthwise_conv(x, self.dw), self.pw, self.pb, 2 * C)
        u = ops.layer_norm(u, self.gln_weight, self.gln_bias, 1e-12)
        cx = x + u[..., :C] * torch.sigmoid(u[..., C:])
        if rowmask is not None:
            cx = cx * rowmask.view(B, T, 1)
        y = ops.layer_norm(lx + cx, self.ln_weight, self.ln_bias, 1e-5)
        if self.training and self.drop_prob > 0.0:                               # DropPath per utterance (convnext.py:106-129)
            from .. import kernels as K
            sc, _ = K.drop_path_rows([self.drop_prob], None, B, T, rng.seed(), self._drop_stream, x.device)
            y = y * sc.view(B, T, 1)
        return x + y


class LeanSpeechBackbone(nn.Module):
    """leanspeech.py:14-39.  forward(x (B, T, C), padding_mask (B, T) True = pad) -> (B, T, C)."""

    def __init__(self, dim, kernel_size, num_layers, drop_path=0.0, conv_layer_cls=None):
        super().__init__()
        assert conv_layer_cls is None, "only the default ConvSeparable conv layer is built"
        rates = [r.item() for r in torch.linspace(0, drop_path, num_layers)]
        self.layers = nn.ModuleList([LeanSpeechBlock(dim, kernel_size, r) for r in rates])

    def forward(self, x, padding_mask):
        rm = row_mask(padding_mask)
        for layer in self.layers:
            x = layer(x, rm)
        return x
