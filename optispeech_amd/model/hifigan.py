"""Host-side mirror of optispeech/model/vocoder/streaming_hifigan: the causal
HiFi-GAN generator -- CausalConv1d / CausalConvTranspose1d (modules/conv_layer.py:118-200), HiFiGANResidualBlock
(modules/residual_block.py:24-106), MultiReceptiveField (modules/multi_fusion.py:24-80), Generator / StreamGenerator
(__init__.py:28-230) -- with the reference's constructor arguments and weight-norm state-dict keys
(``input_conv.conv.{weight_g,weight_v,bias}``, ``upsamples.i.deconv.*``, ``blocks.i.blocks.j.convs1.m.conv.*``, ``pad_buffer``).

The reference module is dead code upstream (its package __init__ raises ImportError); north_star nevertheless
names its dilated and transposed convolutions as hand-written kernels.  Every convolution here is an osp_conv1d_dilated_* /
osp_conv_transpose1d_* call (csrc/a16_conv.hip) on channels-last frames, weight norm is osp_wnorm_fwd / osp_wnorm_bwd; the
element-wise pieces between them (LeakyReLU, residual add, tanh, the replication pad of the transposed conv) are torch glue.
Layout: this module keeps the reference's (B, C, T) at its boundary and runs channels-last inside.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import kernels as K
from .. import precision
from .._lib import call
from ..ops import gsink, _want


def _isbf(t):
    return int(t is not None and t.dtype == torch.bfloat16)


def _split(x32):
    """x = hi + lo in bf16 halves (|lo| <= 2^-9 |x|): hi*hi + lo*hi + hi*lo reproduces the f32 product to ~2e-5 relative."""
    hi = K.cast_bf16(x32.contiguous())
    return hi, K.cast_bf16(x32 - hi.float())


def _terms(a32, b32):
    """Operand pairs of one product: a single bf16 x bf16 call in bf16 mode, three split calls in the f32 parity mode."""
    if precision.is_bf16():
        return [(a32, b32)]
    ah, al = _split(a32)
    bh, bl = _split(b32)
    return [(ah, bh), (al, bh), (ah, bl)]


# ------------------------------------------------------------------------------------------------ kernels (C ABI)
def conv1d_dilated_fwd(x, w, bias, k, dil, pad_left):
    """x (B,T,Cin) f32, w native (Cout,k,Cin) f32 -> (B,T,Cout) f32."""
    B, T, Cin = x.shape
    Cout = w.shape[0]
    y = torch.empty((B, T, Cout), device=x.device, dtype=torch.float32)
    for i, (a, b) in enumerate(_terms(x.contiguous(), w.contiguous())):
        call("osp_conv1d_dilated_fwd", a, _isbf(a), b, _isbf(b), bias if i == 0 else None, y, 0, B, T, Cin, Cout, k, dil, pad_left,
             int(i > 0))
    return y


def conv1d_dilated_bwd(dy, x, w, k, dil, pad_left, need_dx, need_dw):
    """-> (dx (B,T,Cin) or None, dw native (Cout,k,Cin) or None, db (Cout,) or None), all f32."""
    B, T, Cin = x.shape
    Cout = w.shape[0]
    dy = dy.contiguous()
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.zeros((Cout, k, Cin), device=x.device, dtype=torch.float32) if need_dw else None
    db = torch.zeros((Cout,), device=x.device, dtype=torch.float32) if need_dw else None
    if need_dx:
        for i, (a, b) in enumerate(_terms(dy, w.contiguous())):
            call("osp_conv1d_dilated_bwd", a, _isbf(a), None, 0, b, _isbf(b), dx, 0, None, None, B, T, Cin, Cout, k, dil, pad_left,
                 int(i > 0))
    if need_dw:
        for i, (a, b) in enumerate(_terms(dy, x.contiguous())):
            call("osp_conv1d_dilated_bwd", a, _isbf(a), b, _isbf(b), None, 0, None, 0, dw, db if i < 2 else None, B, T, Cin, Cout, k, dil,
                 pad_left, 0)
    return dx, dw, db


def conv_transpose1d_fwd(x, wt, bias, k, s):
    """x (B,T,Cin) f32, wt (Cin,k,Cout) f32 (the transposed conv's own layout) -> (B,(T-1)*s+k,Cout) f32."""
    B, T, Cin = x.shape
    Cout = wt.shape[2]
    wn = wt.permute(2, 1, 0).contiguous()                      # (Cout, k, Cin): the forward reads the reduction index contiguously
    y = torch.empty((B, (T - 1) * s + k, Cout), device=x.device, dtype=torch.float32)
    for i, (a, b) in enumerate(_terms(x.contiguous(), wn)):
        call("osp_conv_transpose1d_fwd", a, _isbf(a), b, _isbf(b), bias if i == 0 else None, y, 0, B, T, Cin, Cout, k, s, int(i > 0))
    return y


def conv_transpose1d_bwd(dy, x, wt, k, s, need_dx, need_dw):
    """-> (dx (B,T,Cin) or None, dwt (Cin,k,Cout) or None, db (Cout,) or None)."""
    B, T, Cin = x.shape
    Cout = wt.shape[2]
    dy = dy.contiguous()
    dx = torch.empty_like(x) if need_dx else None
    dwt = torch.zeros((Cin, k, Cout), device=x.device, dtype=torch.float32) if need_dw else None
    db = None
    if need_dx:
        for i, (a, b) in enumerate(_terms(dy, wt.contiguous())):
            call("osp_conv_transpose1d_bwd", a, _isbf(a), None, 0, b, _isbf(b), dx, 0, None, B, T, Cin, Cout, k, s, int(i > 0))
    if need_dw:
        for a, b in _terms(dy, x.contiguous()):
            call("osp_conv_transpose1d_bwd", a, _isbf(a), b, _isbf(b), None, 0, None, 0, dwt, B, T, Cin, Cout, k, s, 0)
        if Cout % 4 == 0:
            db = torch.zeros((Cout,), device=x.device, dtype=torch.float32)
            call("osp_colsum_prod", dy.view(-1, Cout), None, None, db, dy.numel() // Cout, Cout)
        else:
            db = dy.sum((0, 1))
    return dx, dwt, db


# ------------------------------------------------------------------------------------------------ autograd pairs
class _WNConv1dFn(torch.autograd.Function):
    """conv1d(x, g * v / ||v||, bias) with dilation and a left pad, channels-last; v (Cout,Cin,k), g (Cout,1,1) reference layout."""

    @staticmethod
    def forward(ctx, x, v, g, b, dil, pad_left):
        Cout, Cin, k = v.shape
        _, wn32, _, inv = K.wnorm_fwd(v.detach().view(Cout, Cin, k, 1), g.detach().view(Cout, 1, 1, 1), want_f32=True, want_t=False)
        w = wn32.view(Cout, k, Cin)
        y = conv1d_dilated_fwd(x, w, b.detach() if b is not None else None, k, dil, pad_left)
        ctx.save_for_backward(x, w, inv)
        ctx.params, ctx.cfg = (v, g, b), (k, dil, pad_left)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, inv = ctx.saved_tensors
        v, g, b = ctx.params
        k, dil, pad_left = ctx.cfg
        Cout, Cin, _ = v.shape
        need_w = _want(v) or _want(g) or _want(b)
        dx, dw, db = conv1d_dilated_bwd(dy, x, w, k, dil, pad_left, ctx.needs_input_grad[0], need_w)
        if need_w:
            K.wnorm_bwd(dw.view(Cout, 1, k, Cin), v.detach().view(Cout, Cin, k, 1), g.detach().view(Cout, 1, 1, 1), inv,
                        gsink(v).view(Cout, Cin, k, 1), gsink(g).view(Cout, 1, 1, 1))
            if b is not None:
                gsink(b).add_(db)
        return dx, None, None, None, None, None


class _WNConvTranspose1dFn(torch.autograd.Function):
    """conv_transpose1d(x, g * v / ||v||, bias, stride s); v (Cin,Cout,k), g (Cin,1,1): torch's weight_norm(dim=0) normalises per
    INPUT channel here, which is the leading index of the (Cin, k, Cout) layout the transposed conv runs on."""

    @staticmethod
    def forward(ctx, x, v, g, b, s):
        Cin, Cout, k = v.shape
        _, wn32, _, inv = K.wnorm_fwd(v.detach().view(Cin, Cout, k, 1), g.detach().view(Cin, 1, 1, 1), want_f32=True, want_t=False)
        wt = wn32.view(Cin, k, Cout)
        y = conv_transpose1d_fwd(x, wt, b.detach() if b is not None else None, k, s)
        ctx.save_for_backward(x, wt, inv)
        ctx.params, ctx.cfg = (v, g, b), (k, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wt, inv = ctx.saved_tensors
        v, g, b = ctx.params
        k, s = ctx.cfg
        Cin, Cout, _ = v.shape
        need_w = _want(v) or _want(g) or _want(b)
        dx, dwt, db = conv_transpose1d_bwd(dy, x, wt, k, s, ctx.needs_input_grad[0], need_w)
        if need_w:
            K.wnorm_bwd(dwt.view(Cin, 1, k, Cout), v.detach().view(Cin, Cout, k, 1), g.detach().view(Cin, 1, 1, 1), inv,
                        gsink(v).view(Cin, Cout, k, 1), gsink(g).view(Cin, 1, 1, 1))
            if b is not None:
                gsink(b).add_(db)
        return dx, None, None, None, None


def _batch_buf(buf, B):
    """The streaming pad buffer for a batch of B (the reference streams one utterance; a fresh buffer is (1, C, pad))."""
    return buf if buf.shape[0] == B else buf[:1].expand(B, -1, -1)


# ------------------------------------------------------------------------------------------------ modules (reference contracts)
class _Conv(nn.Module):
    """``conv`` / ``deconv`` child holding weight_g / weight_v / bias exactly as torch.nn.utils.weight_norm names them."""

    def __init__(self, weight, bias):
        super().__init__()
        self.weight_g = nn.Parameter(weight.flatten(1).norm(dim=1).view(-1, 1, 1))
        self.weight_v = nn.Parameter(weight)
        self.bias = nn.Parameter(torch.zeros(bias)) if bias else None


class CausalConv1d(nn.Module):
    """conv_layer.py:118-159.  forward(x (B, C, T)) -> (B, C', T): left zero pad (k-1)*dilation, conv1d(dilation)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=True, pad_buffer=None):
        super().__init__()
        if stride != 1 or groups != 1:
            raise ValueError("the HiFi-GAN generator path uses stride 1, groups 1 (MultiGroupConv1d is not built)")
        ref = nn.Conv1d(in_channels, out_channels, kernel_size, dilation=dilation, bias=bias)       # reference default init
        self.conv = _Conv(ref.weight.detach().clone(), out_channels if bias else 0)
        if bias:
            with torch.no_grad():
                self.conv.bias.copy_(ref.bias)
        self.kernel_size, self.dilation, self.stride = kernel_size, dilation, stride
        self.pad_length = (kernel_size - 1) * dilation
        self.register_buffer("pad_buffer", pad_buffer if pad_buffer is not None else torch.zeros(1, in_channels, self.pad_length))

    def _run(self, x_cl, pad_left):
        c = self.conv
        return _WNConv1dFn.apply(x_cl, c.weight_v, c.weight_g, c.bias, self.dilation, pad_left)

    def forward(self, x):
        return self._run(x.transpose(1, 2).contiguous(), self.pad_length).transpose(1, 2)

    def forward_cl(self, x_cl):
        """channels-last entry used inside the generator (no layout round trips between layers)."""
        return self._run(x_cl, self.pad_length)

    def inference(self, x):
        """Streaming step: the last pad_length frames of the previous chunk stand in for the zero pad (conv_layer.py:151-154)."""
        x = torch.cat((_batch_buf(self.pad_buffer, x.shape[0]), x), -1)
        if self.pad_length:
            self.pad_buffer = x[:, :, x.shape[-1] - self.pad_length:].detach().clone()
        y = self._run(x.transpose(1, 2).contiguous(), self.pad_length).transpose(1, 2)
        return y[:, :, self.pad_length:]

    def reset_buffer(self):
        self.pad_buffer.zero_()


class CausalConvTranspose1d(nn.Module):
    """conv_layer.py:162-200.  forward(x (B, C, T)) -> (B, C', T*stride): left replication pad ceil(k/s)-1, conv_transpose1d,
    crop [s:-s]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias=True, pad_buffer=None):
        super().__init__()
        ref = nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride, bias=bias)
        self.deconv = _Conv(ref.weight.detach().clone(), out_channels if bias else 0)
        if bias:
            with torch.no_grad():
                self.deconv.bias.copy_(ref.bias)
        self.kernel_size, self.stride = kernel_size, stride
        self.pad_length = math.ceil(kernel_size / stride) - 1
        self.register_buffer("pad_buffer", pad_buffer if pad_buffer is not None else torch.zeros(1, in_channels, self.pad_length))

    def _run(self, x_cl):
        d = self.deconv
        y = _WNConvTranspose1dFn.apply(x_cl, d.weight_v, d.weight_g, d.bias, self.stride)
        return y[:, self.stride: y.shape[1] - self.stride]

    def forward_cl(self, x_cl):
        if self.pad_length:
            x_cl = torch.cat((x_cl[:, :1].expand(-1, self.pad_length, -1), x_cl), 1)        # ReplicationPad1d((pad, 0))
        return self._run(x_cl.contiguous())

    def forward(self, x):
        return self.forward_cl(x.transpose(1, 2)).transpose(1, 2)

    def inference(self, x):
        x = torch.cat((_batch_buf(self.pad_buffer, x.shape[0]), x), -1)
        if self.pad_length:
            self.pad_buffer = x[:, :, x.shape[-1] - self.pad_length:].detach().clone()
        return self._run(x.transpose(1, 2).contiguous()).transpose(1, 2)

    def reset_buffer(self):
        self.pad_buffer.zero_()


class HiFiGANResidualBlock(nn.Module):
    """residual_block.py:24-106: for each dilation  x = x + conv2(act(conv1_dilated(act(x))))."""

    def __init__(self, kernel_size=3, channels=512, dilations=(1, 3, 5), groups=1, bias=True, use_additional_convs=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1}):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        self.use_additional_convs = use_additional_convs
        self.activation = getattr(nn, nonlinear_activation)(**nonlinear_activation_params)
        self.convs1 = nn.ModuleList([CausalConv1d(channels, channels, kernel_size, 1, d, groups, bias) for d in dilations])
        if use_additional_convs:
            self.convs2 = nn.ModuleList([CausalConv1d(channels, channels, kernel_size, 1, 1, groups, bias) for _ in dilations])
        self.num_layer = len(self.convs1)

    def _pass(self, x, name):
        for idx in range(self.num_layer):
            xt = getattr(self.convs1[idx], name)(self.activation(x))
            if self.use_additional_convs:
                xt = getattr(self.convs2[idx], name)(self.activation(xt))
            x = xt + x
        return x

    def forward(self, x):
        return self._pass(x, "forward")

    def forward_cl(self, x):
        return self._pass(x, "forward_cl")

    def inference(self, x):
        return self._pass(x, "inference")


class MultiReceptiveField(nn.Module):
    """multi_fusion.py:24-80: mean of the residual blocks of different kernel sizes."""

    def __init__(self, channels=512, resblock_kernel_sizes=(3, 7, 11), resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)), groups=1,
                 bias=True, use_additional_convs=True, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.1}):
        super().__init__()
        assert len(resblock_kernel_sizes) == len(resblock_dilations)
        self.num_blocks = len(resblock_kernel_sizes)
        self.blocks = nn.ModuleList([HiFiGANResidualBlock(k, channels, d, groups, bias, use_additional_convs, nonlinear_activation,
                                                          nonlinear_activation_params)
                                     for k, d in zip(resblock_kernel_sizes, resblock_dilations)])

    def _pass(self, c, name):
        cs = 0.0
        for blk in self.blocks:
            cs = cs + getattr(blk, name)(c)
        return cs / self.num_blocks

    def forward(self, c):
        return self._pass(c, "forward")

    def forward_cl(self, c):
        return self._pass(c, "forward_cl")

    def inference(self, c):
        return self._pass(c, "inference")


class Generator(nn.Module):
    """streaming_hifigan/__init__.py:28-161 (weight norm on every conv, N(0, 0.01) init).  forward(c (B, in_channels, T)) ->
    (B, out_channels, T * prod(upsample_scales)); ``inference`` is the chunked streaming form of StreamGenerator (:230-)."""

    def __init__(self, in_channels=80, out_channels=1, channels=512, kernel_size=7, upsample_scales=(8, 8, 2, 2),
                 upsample_kernel_sizes=(16, 16, 4, 4), resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)), groups=1, bias=True, use_additional_convs=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1}, use_weight_norm=True,
                 stats=None):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        assert len(upsample_scales) == len(upsample_kernel_sizes) and len(resblock_dilations) == len(resblock_kernel_sizes)
        if not use_weight_norm or stats is not None or groups != 1:
            raise ValueError("built: the weight-normed MultiReceptiveField generator without input statistics")
        self.num_upsamples = len(upsample_kernel_sizes)
        self.input_conv = CausalConv1d(in_channels, channels, kernel_size, stride=1)
        self.upsamples, self.blocks = nn.ModuleList(), nn.ModuleList()
        self.activation_upsamples = getattr(nn, nonlinear_activation)(**nonlinear_activation_params)
        ch = channels
        for i, (s, k) in enumerate(zip(upsample_scales, upsample_kernel_sizes)):
            assert k == 2 * s
            self.upsamples.append(CausalConvTranspose1d(channels // (2 ** i), channels // (2 ** (i + 1)), kernel_size=k, stride=s))
            ch = channels // (2 ** (i + 1))
            self.blocks.append(MultiReceptiveField(ch, resblock_kernel_sizes, resblock_dilations, groups, bias, use_additional_convs,
                                                   nonlinear_activation, nonlinear_activation_params))
        self.activation_output1 = nn.LeakyReLU()
        self.activation_output2 = nn.Tanh()
        self.output_conv = CausalConv1d(ch, out_channels, kernel_size, stride=1)
        self.norm = False
        self.reset_parameters()

    def reset_parameters(self):
        """weight ~ N(0, 0.01) (the official HiFi-GAN initialisation, :163-176); g follows the new norms."""
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, _Conv):
                    m.weight_v.normal_(0.0, 0.01)
                    m.weight_g.copy_(m.weight_v.flatten(1).norm(dim=1).view(-1, 1, 1))

    def forward(self, c):
        h = self.input_conv.forward_cl(c.transpose(1, 2).contiguous())
        for i in range(self.num_upsamples):
            h = self.upsamples[i].forward_cl(self.activation_upsamples(h))
            h = self.blocks[i].forward_cl(h)
        h = self.output_conv.forward_cl(self.activation_output1(h))
        return self.activation_output2(h).transpose(1, 2)

    @torch.no_grad()
    def inference(self, c):
        """One streaming chunk (B, in_channels, t) -> (B, out_channels, t * hop); call reset_buffer() between utterances."""
        c = self.input_conv.inference(c)
        for i in range(self.num_upsamples):
            c = self.upsamples[i].inference(self.activation_upsamples(c))
            c = self.blocks[i].inference(c)
        c = self.output_conv.inference(self.activation_output1(c))
        return self.activation_output2(c)

    def reset_buffer(self):
        for m in self.modules():
            if m is not self and hasattr(m, "reset_buffer"):
                m.reset_buffer()


StreamGenerator = Generator
