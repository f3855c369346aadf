"""autograd glue: each Function is a forward/backward *pair of HIP kernel sequences* behind the C ABI.

Parameter gradients are accumulated by the kernels directly into ``param.grad`` (the flat gradient
arena owned by the trainer, zeroed once per step); the Functions therefore return ``None`` for
parameter inputs.  Only activations flow through autograd.
"""
import torch

from . import kernels as K


def gsink(p):
    """Gradient accumulation target of a parameter (allocated zero on first use)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def _want(p):
    return p is not None and p.requires_grad


class ConvNeXtBlockFn(torch.autograd.Function):
    """ConvNeXtBlock.forward + the backbone's per-block mask (generator/modules/convnext.py:34-47, :99-101).

    y = (x + rowscale * gamma * (W2 gelu(W1 LN(dwconv7(x)) + b1) + b2)) * rowmask
    x (B,T,C); dw (7,C) native tap-major; W1 (I,C); W2 (C,I); rowmask/rowscale (B*T,) or None.
    """

    @staticmethod
    def forward(ctx, x, dw, dwb, lnw, lnb, W1, b1, W2, b2, gamma, rowmask, rowscale):
        B, T, C = x.shape
        I = W1.shape[0]
        M = B * T
        save = any(ctx.needs_input_grad)
        x = x.contiguous()
        h, xhat, rstd = K.dwconv7_ln_fwd(x, dw, dwb, lnw, lnb, 1e-6, save)
        h2 = h.view(M, C)
        u = torch.empty((M, I), device=x.device, dtype=torch.float32) if save else None
        g = K.conv_gemm(h2, W1, I, epi=K.EPI_GELU, bias=b1, aux_out=u)
        z = torch.empty((M, C), device=x.device, dtype=torch.float32) if save else None
        y = K.conv_gemm(g, W2, C, epi=K.EPI_SCALE_RES_MASK, bias=b2, gamma=gamma, res=x.view(M, C),
                        rowmask=rowmask, rowscale=rowscale, aux_out=z)
        if save:
            if rowmask is not None and rowscale is not None:
                rowf = rowmask * rowscale
            else:
                rowf = rowmask if rowmask is not None else rowscale
            ctx.save_for_backward(x, xhat, rstd, h, u, g, z, rowmask, rowf)
            ctx.params = (dw, dwb, lnw, lnb, W1, b1, W2, b2, gamma)
        return y.view(B, T, C)

    @staticmethod
    def backward(ctx, dy):
        x, xhat, rstd, h, u, g, z, rowmask, rowf = ctx.saved_tensors
        dw, dwb, lnw, lnb, W1, b1, W2, b2, gamma = ctx.params
        B, T, C = x.shape
        I = W1.shape[0]
        M = B * T
        dy2 = dy.contiguous().view(M, C)
        if _want(gamma):
            t = dy2 * z
            if rowf is not None:
                t = t * rowf[:, None]
            gsink(gamma).add_(t.sum(0))
        W2g = W2 * gamma[:, None]
        # du[m,k] = rowf[m] * sum_n dy[m,n] * gamma[n] W2[n,k] * gelu'(u[m,k])
        du = K.conv_gemm(dy2, W2g, I, cin=C, w_strides=(1, 0, I), epi=K.EPI_GELU_BWD, rowscale=rowf, aux_in=u)
        if _want(W2):
            K.conv_wgrad(dy2, g, gsink(W2), gsink(b2) if _want(b2) else None, arow=rowf, oscale=gamma)
        dh = K.conv_gemm(du, W1, C, cin=I, w_strides=(1, 0, C))
        if _want(W1):
            K.conv_wgrad(du, h.view(M, C), gsink(W1), gsink(b1) if _want(b1) else None)
        wl = _want(lnw)
        dc = K.layernorm_bwd(dh, xhat.view(M, C), None, rstd.view(M), lnw, gsink(lnw) if wl else None,
                             gsink(lnb) if wl else None)
        wd = _want(dw)
        dx = K.dwconv7_bwd(dc.view(B, T, C), x, dw, dy2.view(B, T, C), rowmask, gsink(dw) if wd else None,
                           gsink(dwb) if wd else None)
        return (dx,) + (None,) * 11


class LayerNormFn(torch.autograd.Function):
    """y = (LN_C(x) * w + b) * dropout * rowmask  (nn.LayerNorm call sites convnext.py:102, wavenext:84)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, rowmask, drop_p, seed, stream_id):
        x = x.contiguous()
        save = any(ctx.needs_input_grad)
        y, mean, rstd = K.layernorm_fwd(x, w, b, eps, save=save, rowmask=rowmask, drop_p=drop_p, seed=seed,
                                        stream_id=stream_id)
        if save:
            ctx.save_for_backward(x, mean, rstd, rowmask)
            ctx.params = (w, b)
            ctx.cfg = (drop_p, seed, stream_id)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, rowmask = ctx.saved_tensors
        w, b = ctx.params
        drop_p, seed, stream_id = ctx.cfg
        ww = _want(w)
        dx = K.layernorm_bwd(dy.contiguous(), x, mean, rstd, w, gsink(w) if ww else None, gsink(b) if ww else None,
                             rowmask=rowmask, drop_p=drop_p, seed=seed, stream_id=stream_id)
        return (dx,) + (None,) * 7


def layer_norm(x, w, b, eps, rowmask=None, drop_p=0.0, seed=0, stream_id=0):
    return LayerNormFn.apply(x, w, b, eps, rowmask, drop_p, seed, stream_id)
