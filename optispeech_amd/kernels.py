"""Thin, shape-checked Python wrappers over the C-ABI entry points (one function per kernel family).

Everything here is channels-last fp32 on the GPU; outputs are allocated with torch (device memory
plumbing only).  No arithmetic happens in Python.
"""
import torch

from ._lib import call

EPI_NONE, EPI_RELU, EPI_GELU, EPI_SCALE_RES_MASK, EPI_GELU_BWD, EPI_RELU_BWD, EPI_AXMY, EPI_MASK = range(8)


def _f32(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.float32 and t.is_cuda, (t.dtype, t.device)


def conv_gemm(a, w, n_out, *, T=None, taps=1, pad=0, cin=None, w_strides=None, out=None, epi=EPI_NONE, bias=None,
              gamma=None, res=None, rowmask=None, rowscale=None, aux_out=None, aux_in=None, a_rowscale=None,
              accumulate=False, batch=1, batch_strides=(0, 0, 0, 0), ldc=None, lda=None):
    """C[m, n] = epi(sum_{j,c} A[m + j - pad, c] * W(n, j, c)).

    a: (M, Cin) view (row stride lda), M = utterances * T.  w: weight tensor; ``w_strides`` =
    (stride_n, stride_tap, stride_k) in elements; default is the native (N, taps, Cin) layout.
    """
    _f32(a, w, bias, gamma, res, rowmask, rowscale, aux_out, aux_in, a_rowscale, out)
    M = a.shape[-2]
    cin = a.shape[-1] if cin is None else cin
    lda = a.stride(-2) if lda is None else lda
    T = M if T is None else T
    if w_strides is None:
        w_strides = (taps * cin, cin, 1)
    if out is None:
        shape = (batch, M, n_out) if batch > 1 else (M, n_out)
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
    ldc = out.stride(-2) if ldc is None else ldc
    ld_aux = 0
    for t in (aux_out, aux_in):
        if t is not None:
            ld_aux = t.stride(-2)
    ldr = res.stride(-2) if res is not None else 0
    call("osp_conv_gemm_f32", a, lda, M, T, cin, taps, pad, a_rowscale, w, w_strides[0], w_strides[1], w_strides[2],
         n_out, out, ldc, epi, bias, gamma, res, ldr, rowmask, rowscale, aux_out, aux_in, ld_aux, batch,
         batch_strides[0], batch_strides[1], batch_strides[2], batch_strides[3], bool(accumulate))
    return out


def conv_wgrad(dy, x, dw, db=None, *, T=None, taps=1, pad=0, arow=None, oscale=None):
    """dw[n, j, c] += oscale[n] * sum_m arow[m] dy[m, n] x[m + j - pad, c];  db[n] += oscale[n] * sum_m arow[m] dy[m, n]."""
    _f32(dy, x, dw, db, arow, oscale)
    M, N = dy.shape[-2], dy.shape[-1]
    cin = x.shape[-1]
    T = M if T is None else T
    assert dw.is_contiguous() and dw.numel() == N * taps * cin, (dw.shape, N, taps, cin)
    call("osp_conv_wgrad_f32", dy, dy.stride(-2), x, x.stride(-2), M, T, N, cin, taps, pad, arow, oscale, dw,
         taps * cin, db)


def dwconv7_ln_fwd(x, dw, dwb, lnw, lnb, eps, save):
    _f32(x, dw, dwb, lnw, lnb)
    B, T, C = x.shape
    assert x.is_contiguous() and dw.shape == (7, C)
    h = torch.empty_like(x)
    xhat = torch.empty_like(x) if save else None
    rstd = torch.empty((B, T), device=x.device, dtype=torch.float32) if save else None
    call("osp_dwconv7_ln_fwd", x, dw, dwb, lnw, lnb, float(eps), h, xhat, rstd, B, T, C)
    return h, xhat, rstd


def dwconv7_bwd(dc, x, dw, dres, dres_rowmask, ddw, ddb):
    _f32(dc, x, dw, dres, dres_rowmask, ddw, ddb)
    B, T, C = x.shape
    dx = torch.empty_like(x)
    call("osp_dwconv7_bwd", dc, x, dw, dres, dres_rowmask, dx, ddw, ddb, B, T, C)
    return dx


def layernorm_fwd(x, w, b, eps, *, save=True, rowmask=None, drop_p=0.0, seed=0, stream_id=0):
    _f32(x, w, b, rowmask)
    C = x.shape[-1]
    rows = x.numel() // C
    assert x.is_contiguous()
    y = torch.empty_like(x)
    mean = torch.empty((rows,), device=x.device, dtype=torch.float32) if save else None
    rstd = torch.empty((rows,), device=x.device, dtype=torch.float32) if save else None
    call("osp_layernorm_fwd", x, w, b, float(eps), y, mean, rstd, rowmask, float(drop_p), int(seed), int(stream_id),
         rows, C)
    return y, mean, rstd


def layernorm_bwd(dy, xin, mean, rstd, w, dlnw, dlnb, *, relu_src=None, rowmask=None, drop_p=0.0, seed=0,
                  stream_id=0):
    _f32(dy, xin, mean, rstd, w, relu_src, rowmask, dlnw, dlnb)
    C = dy.shape[-1]
    rows = dy.numel() // C
    assert dy.is_contiguous() and xin.is_contiguous()
    dx = torch.empty_like(dy)
    call("osp_layernorm_bwd", dy, xin, mean, rstd, w, relu_src, rowmask, float(drop_p), int(seed), int(stream_id),
         dx, dlnw, dlnb, rows, C)
    return dx
