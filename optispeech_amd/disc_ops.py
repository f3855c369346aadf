"""Multi-period discriminator stacks on the bf16-MFMA conv-GEMM.

DiscriminatorP (vocoder/wavenext/disc/_discriminators.py:41-97) applies Conv2d((k,1), stride (s,1)) over the
period-folded wave (B, 1, T/p, p): every (utterance, period column) is an independent 1-D sequence.  In channels-last
layout (U = B*p sequences, T frames, C channels) each layer is a strided k-tap Conv1d = one GEMM, LeakyReLU fused in
the epilogue, activations kept in bf16.  The whole stack is ONE autograd Function so the backward can fuse
(dgrad + incoming feature-matching gradient) * LeakyReLU' into each dgrad epilogue.
"""
import torch

from . import kernels as K


def _tout(tin, taps, stride, pad):
    return (tin + 2 * pad - taps) // stride + 1


def conv1d_strided_fwd(x, w, bias, taps, stride, pad, slope, out_bf16):
    """x (U,Tin,Cin) f32|bf16; w native (Cout,taps,Cin) f32|bf16 -> (U,Tout,Cout)."""
    U, Tin, Cin = x.shape
    Cout = w.shape[0]
    Tout = _tout(Tin, taps, stride, pad)
    y = K.conv_gemm_bf16(x.view(U * Tin, Cin), w, Cout, M=U * Tout, Trows=Tout, Tin=Tin, cin=Cin, taps=taps,
                         a_step=stride, a_off=-pad, bias=bias, epi=K.EPI_LRELU if slope is not None else K.EPI_NONE,
                         slope=slope or 0.0, out_bf16=out_bf16)
    return y.view(U, Tout, Cout)


def transpose_weight(w):
    """native (Cout, taps, Cin) -> dgrad layout (Cin, taps, Cout) in bf16: the reduction index of the dgrad GEMM
    (tap, n) becomes contiguous, so dgrad runs on the same fast k-contiguous loader as the forward."""
    return K.cast_bf16(w.float().permute(2, 1, 0).contiguous()) if w.dtype != torch.bfloat16 else w.permute(2, 1, 0).contiguous()


def conv1d_strided_dgrad(dy, w, Tin, Cin, taps, stride, pad, *, lrelu_y=None, extra=None, slope=0.1, out_bf16=False,
                         wt=None):
    """dx (U,Tin,Cin) of a strided conv.  One GEMM per phase r = t_in mod stride (only the taps j = (r+pad) mod stride
    + stride*i contribute), so no multiply-by-zero work.  Optional fused epilogue: (dx + extra) * lrelu'(lrelu_y).
    ``wt`` = transpose_weight(w) (computed here when not supplied)."""
    U, Tout, Cout = dy.shape
    if wt is None:
        wt = transpose_weight(w)
    dx = torch.empty((U, Tin, Cin), device=dy.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    dy2 = dy.view(U * Tout, Cout)
    epi = K.EPI_LRELU_BWD if lrelu_y is not None else K.EPI_NONE
    covered = 0
    for r in range(stride):
        q_r = (Tin - r + stride - 1) // stride
        if q_r <= 0:
            continue
        j0 = (r + pad) % stride
        n_i = (taps - j0 + stride - 1) // stride if j0 < taps else 0
        if n_i == 0:
            dx[:, r::stride].zero_()
            continue
        off = (r + pad - j0) // stride
        base = wt.view(Cin, taps, Cout)[:, j0:, :]                      # pointer to tap j0 of the (Cin, taps, Cout) copy
        K.conv_gemm_bf16(dy2, base, Cin, M=U * q_r, Trows=q_r, Tin=Tout, cin=Cout, taps=n_i, a_step=1, a_tapstep=-1,
                         a_off=off, w_strides=(taps * Cout, stride * Cout, 1), out=dx.view(U * Tin, Cin), ldc=Cin, Tc=Tin,
                         c_step=stride, c_off=r, epi=epi, aux_in=None if lrelu_y is None else lrelu_y.view(U * Tin, Cin),
                         res=None if extra is None else extra.view(U * Tin, Cin), slope=slope)
        covered += q_r
    assert covered == Tin or stride > Tin
    return dx


def conv1d_strided_bwd(dy, x, w, dw, db, taps, stride, pad, need_dx):
    """Test/helper entry: dgrad (+ wgrad accumulated into dw/db) of y = conv1d(x, w, stride, pad)."""
    U, Tin, Cin = x.shape
    Tout, Cout = dy.shape[1], dy.shape[2]
    K.conv_wgrad_bf16(dy.view(U * Tout, Cout), x.view(U * Tin, Cin), dw, db, M=U * Tout, Trows=Tout, Tin=Tin, n=Cout,
                      cin=Cin, taps=taps, pad=pad, x_step=stride)
    return conv1d_strided_dgrad(dy, w, Tin, Cin, taps, stride, pad) if need_dx else None


class MPDStackFn(torch.autograd.Function):
    """The six convolutions of one DiscriminatorP on a batch of period-folded sequences.

    inputs : x (U, T0, 1) f32; then (w_i native (Cout, taps, Cin) f32, b_i) for the 5 convs and conv_post
    outputs: y1..y5 (bf16, LeakyReLU applied; y2..y5 are the reference's fmap entries) and the score s (U, T5, 1) f32
    """
    STRIDES = (3, 3, 3, 3, 1)
    SLOPE = 0.1

    @staticmethod
    def forward(ctx, x, *wb):
        ctx.set_materialize_grads(False)
        ws, bs = wb[0::2], wb[1::2]
        acts, h = [], x.contiguous()
        wbf = [K.cast_bf16(w) for w in ws]
        ctx.need_dgrad = any(ctx.needs_input_grad)
        for i in range(5):
            h = conv1d_strided_fwd(h, wbf[i], bs[i], 5, MPDStackFn.STRIDES[i], 2, MPDStackFn.SLOPE, True)
            acts.append(h)
        s = conv1d_strided_fwd(h, wbf[5], bs[5], 3, 1, 1, None, False)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, *acts, *wbf)
            ctx.wneed = [w.requires_grad for w in ws]
            ctx.bneed = [b.requires_grad for b in bs]
        return (*acts, s)

    @staticmethod
    def backward(ctx, d1, d2, d3, d4, d5, ds):
        saved = ctx.saved_tensors
        x, acts, wbf = saved[0], saved[1:6], saved[6:12]
        extras = [d.contiguous() if d is not None else None for d in (d1, d2, d3, d4, d5)]
        if ds is None:
            ds = torch.zeros((x.shape[0], acts[4].shape[1], 1), device=x.device, dtype=torch.float32)
        U = x.shape[0]
        grads_w, grads_b = [None] * 6, [None] * 6

        def wgrad(i, g, inp, taps, pad, stride):
            if not (ctx.wneed[i] or ctx.bneed[i]):
                return
            Cout, Cin = wbf[i].shape[0], wbf[i].shape[2]
            dw = torch.zeros(wbf[i].shape, device=g.device, dtype=torch.float32)
            db = torch.zeros(Cout, device=g.device, dtype=torch.float32)
            Tin, Tout = inp.shape[1], g.shape[1]
            K.conv_wgrad_bf16(g.view(U * Tout, Cout), inp.reshape(U * Tin, Cin), dw, db, M=U * Tout, Trows=Tout,
                              Tin=Tin, n=Cout, cin=Cin, taps=taps, pad=pad, x_step=stride)
            grads_w[i], grads_b[i] = dw, db

        wts = [transpose_weight(w) for w in wbf]
        # conv_post: s = conv(y5)
        g = ds.contiguous()
        wgrad(5, g, acts[4], 3, 1, 1)
        g = conv1d_strided_dgrad(g, wbf[5], acts[4].shape[1], acts[4].shape[2], 3, 1, 1, lrelu_y=acts[4],
                                 extra=extras[4], slope=MPDStackFn.SLOPE, out_bf16=True, wt=wts[5])
        for i in range(4, -1, -1):
            inp = acts[i - 1] if i > 0 else x
            st = MPDStackFn.STRIDES[i]
            wgrad(i, g, inp, 5, 2, st)
            if i > 0:
                g = conv1d_strided_dgrad(g, wbf[i], inp.shape[1], inp.shape[2], 5, st, 2, lrelu_y=inp,
                                         extra=extras[i - 1], slope=MPDStackFn.SLOPE, out_bf16=True, wt=wts[i])
            elif ctx.needs_input_grad[0]:
                g = conv1d_strided_dgrad(g, wbf[0], inp.shape[1], 1, 5, st, 2, out_bf16=False, wt=wts[0])
            else:
                g = None
        out = [g]
        for i in range(6):
            out += [grads_w[i], grads_b[i]]
        return tuple(out)
