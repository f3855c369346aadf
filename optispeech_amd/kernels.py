"""Thin, shape-checked Python wrappers over the C-ABI entry points (one function per kernel family).

Everything here is channels-last fp32 on the GPU; outputs are allocated with torch (device memory
plumbing only).  No arithmetic happens in Python.
"""
import os

import torch

from . import precision as _precision
from ._lib import call

EPI_NONE, EPI_RELU, EPI_GELU, EPI_SCALE_RES_MASK, EPI_GELU_BWD, EPI_RELU_BWD, EPI_AXMY, EPI_MASK = range(8)


def _keep(*objs):
    """While a call tape is being recorded (optispeech_amd/tape.py): tensors whose addresses travel inside a HOST descriptor table
    (the *_multi entry points) are invisible to the tape's own bookkeeping -- keep them alive as long as the tape."""
    from ._lib import lib
    f = lib()._fast
    if f is not None and f.tape_recording():
        for o in objs:
            if o is not None:
                f.tape_keep(o)


def _persistent(owner, slot, key, shape, dtype):
    """A buffer that keeps its ADDRESS for the lifetime of ``owner`` (a Parameter): derived weight packs are refreshed in place every
    optimizer epoch instead of re-allocated, because the call tapes of the regions that read them hold raw addresses."""
    bufs = getattr(owner, slot, None)
    if bufs is None:
        bufs = {}
        setattr(owner, slot, bufs)
    t = bufs.get(key)
    if t is None or tuple(t.shape) != tuple(shape) or t.device != owner.device or t.dtype != dtype:
        t = bufs[key] = torch.empty(shape, device=owner.device, dtype=dtype)
    return t


def _seed(seed):
    """(host seed, device seed tensor or None): rng.seed() is an int in eager mode and an rng.DeviceSeed while a step is being
    captured into / replayed from a hipGraph (the per-step seed then lives in device memory, csrc: ``seed_dev``)."""
    if isinstance(seed, int):
        return seed, None
    return 0, seed.tensor


def _f32(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.float32 and t.is_cuda, (t.dtype, t.device)


def conv_gemm(a, w, n_out, *, T=None, taps=1, pad=0, cin=None, w_strides=None, out=None, epi=EPI_NONE, bias=None,
              gamma=None, res=None, rowmask=None, rowscale=None, aux_out=None, aux_in=None, a_rowscale=None,
              accumulate=False, batch=1, batch_strides=(0, 0, 0, 0), ldc=None, lda=None, w_param=None):
    """C[m, n] = epi(sum_{j,c} A[m + j - pad, c] * W(n, j, c)).

    a: (M, Cin) view (row stride lda), M = utterances * T.  w: weight tensor; ``w_strides`` =
    (stride_n, stride_tap, stride_k) in elements; default is the native (N, taps, Cin) layout.
    w_param: the leaf Parameter ``w`` is a view of -- its bf16 pack (performance mode) is then kept on the Parameter for the
    current optimizer epoch instead of being re-made (k-strided views) or converted in the kernel's loader (contiguous ones).
    """
    _f32(a, w, bias, gamma, res, rowmask, rowscale, aux_out, aux_in, a_rowscale, out)
    M = a.shape[-2]
    cin = a.shape[-1] if cin is None else cin
    lda = a.stride(-2) if lda is None else lda
    T = M if T is None else T
    if w_strides is None:
        w_strides = (taps * cin, cin, 1)
    if out is None:
        # a batched caller (3-D operand) gets a 3-D result also when its batch happens to be 1 (a batch of one utterance)
        shape = (batch, M, n_out) if (batch > 1 or a.dim() == 3) else (M, n_out)
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
    if _precision.is_bf16() and M * n_out * cin * taps >= (1 << 22) and n_out >= 32:
        # performance mode: same contraction on the bf16 MFMA kernel (f32 storage, converted while staging)
        ldc_ = out.stride(-2) if ldc is None else ldc
        ld_aux_ = 0
        for t in (aux_out, aux_in):
            if t is not None:
                ld_aux_ = t.stride(-2)
        w_bf16 = 0
        if batch == 1 and w_param is not None:
            # bf16 k-contiguous pack, cached per optimizer epoch (forward and flipped-tap dgrad views are separate entries)
            w, w_bf16, w_strides = _param_pack(w_param, w, n_out, taps, cin, w_strides), 1, (taps * cin, cin, 1)
        elif batch == 1 and w_strides[2] != 1:
            # k-strided (transposed / flipped) weights: one pack launch into the k-contiguous bf16 layout
            wp = torch.empty((n_out, taps, cin), device=a.device, dtype=torch.bfloat16)
            call("osp_pack_bf16", w, None, wp, n_out, taps, cin, w_strides[0], w_strides[1], w_strides[2])
            w, w_bf16, w_strides = wp, 1, (taps * cin, cin, 1)
        call("osp_conv_gemm_bf16", a, 0, lda, M, T, T, cin, taps, 1, 1, -pad, a_rowscale, w, w_bf16, w_strides[0], w_strides[1],
             w_strides[2], n_out, out, 0, ldc_, T, 1, 0, epi, bias, gamma, res, 0, res.stride(-2) if res is not None else 0,
             rowmask, rowscale, aux_out, aux_in, 0, ld_aux_, 0.0, batch, batch_strides[0], batch_strides[1],
             batch_strides[2], batch_strides[3], bool(accumulate))
        return out
    if (batch == 1 and w_param is not None and w_strides[2] != 1 and a.is_cuda and -(-M // 128) * -(-n_out // 64) >= 64
            and f32_packs_ok()):
        # exact-f32 modes: a k-strided view of a PARAMETER (the flipped-tap / transposed weights of an input-gradient conv) is
        # packed k-contiguous once per optimizer epoch, so that the GEMM can take the direct-to-LDS kernel (only at sizes that
        # kernel takes)
        w, w_strides = _param_pack_f32(w_param, w, n_out, taps, cin, w_strides), (taps * cin, cin, 1)
    ldc = out.stride(-2) if ldc is None else ldc
    ld_aux = 0
    for t in (aux_out, aux_in):
        if t is not None:
            ld_aux = t.stride(-2)
    ldr = res.stride(-2) if res is not None else 0
    # "mixed" parity mode outside the index-critical path: f32 operands, split-bf16 products (precision.f32_split)
    call("osp_conv_gemm_f32_split" if _precision.f32_split() else "osp_conv_gemm_f32",
         a, lda, M, T, cin, taps, pad, a_rowscale, w, w_strides[0], w_strides[1], w_strides[2],
         n_out, out, ldc, epi, bias, gamma, res, ldr, rowmask, rowscale, aux_out, aux_in, ld_aux, batch,
         batch_strides[0], batch_strides[1], batch_strides[2], batch_strides[3], bool(accumulate))
    return out


def _tape_recording():
    """True while a call tape is being recorded or a hipGraph captured: no torch-made weight pack may be created (a tape would be
    poisoned; a pack allocated during a capture lives in the graph's private pool and must not be cached on the Parameter)."""
    from . import tape
    return tape.recording() or torch.cuda.is_current_stream_capturing()


_F32_PACKS_UNDER_TAPE = __import__('os').environ.get('OSP_F32_PACKS_UNDER_TAPE', '1') != '0'


def f32_packs_ok():
    """f32 weight packs are made by the library (osp_pack_bf16_multi, out_f32 rows) into persistent buffers: a call tape can hold them.
    Not inside a hipGraph capture (a pack refreshed there would be baked into the graph's first replay only)."""
    if _F32_PACKS_UNDER_TAPE:
        return not torch.cuda.is_current_stream_capturing()
    return not _tape_recording()                  # OSP_F32_PACKS_UNDER_TAPE=0: round 4's rule (A/B runs)


def _param_pack_f32(p, w, n_out, taps, cin, w_strides):
    """(n_out, taps, cin) f32 copy of the strided view ``w`` of Parameter ``p`` (element strides ``w_strides``, negative ones allowed),
    cached on ``p`` for the current optimizer epoch: the same registry, buffers and multi launch as the bf16 packs (round 5; a torch
    gather before, which no call tape could hold)."""
    return _param_pack(p, w, n_out, taps, cin, w_strides, f32=True)


def _region_first_use(p, key):
    """True exactly once per (recorded region, pack).  While a call tape is RECORDING, the first use of an epoch-cached weight pack in
    the region counts as a cache miss, whatever the cache says: a region recorded when the pack happened to be fresh --
    the second differently-shaped micro-batch of a gradient-accumulation epoch, a region whose packs another region refreshed --
    would otherwise put no refresh launch on its tape and, replayed as the FIRST user of a later epoch, read the previous epoch's
    weights.  With the rule every tape carries the refresh of every pack it reads; in the steady state (one optimizer step per
    training step) that is exactly the launch an epoch's first use makes anyway."""
    from . import tape
    seen = tape.region_packs()
    if seen is None:
        return False
    k = (id(p), key)
    if k in seen:
        return False
    seen.add(k)
    return True


def _region_mark(p, key):
    from . import tape
    seen = tape.region_packs()
    if seen is not None:
        seen.add((id(p), key))


_PACK_REG = {}          # (id(param), view key) -> (weakref(param), view key): every pack ever asked for through _param_pack


def _param_pack(p, w, n_out, taps, cin, w_strides, f32=False):
    """(n_out, taps, cin) bf16 pack of the view ``w`` of Parameter ``p`` (element strides ``w_strides``), cached on ``p`` with the
    invalidation rule of param_bf16: optimizer epoch, in-place version, storage address.  A miss refreshes EVERY registered pack
    that is stale (all of them after an optimizer step) in one multi launch instead of one launch per weight and view."""
    import weakref
    from . import values
    key = (w.data_ptr() - p.data_ptr(), n_out, taps, cin, tuple(w_strides)) + (("f32",) if f32 else ())
    rk = (id(p), key)
    reg = _PACK_REG.get(rk)
    if reg is None or reg[0]() is not p:        # (id() of a collected Parameter is re-used: an entry with a dead weakref is not p's)
        _PACK_REG[rk] = (weakref.ref(p), key)
    stamp = (values.param_epoch(), p._version, p.data_ptr())
    cache = getattr(p, "_osp_packs", None)
    force = _region_first_use(p, ("view",) + key)            # recording a tape: this region's tape must carry the refresh
    if cache is not None and cache[0] == stamp and not force:
        wp = cache[1].get(key)
        if wp is not None:
            return wp
    if p.is_cuda and torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture: this view only, on the single-tensor kernel (the capture of the segmented step did not survive
        # the batched launch here; the eager step is where the launch count matters)
        if cache is None or cache[0] != stamp:
            cache = (stamp, {})
            p._osp_packs = cache
        assert not f32, "f32 packs are not made inside a hipGraph capture (f32_packs_ok)"
        wp = _persistent(p, "_osp_pack_bufs", key, (n_out, taps, cin), torch.bfloat16)
        call("osp_pack_bf16", w, None, wp, n_out, taps, cin, w_strides[0], w_strides[1], w_strides[2])
        cache[1][key] = wp
        return wp
    import numpy as np
    rows, fills, dead = [], [], []
    epoch = values.param_epoch()
    # only parameters of the SAME flat arena (= the same model's optimizer group) are refreshed together: a captured graph must
    # not read another model's weights (they may be freed while the graph lives); a parameter outside any arena is packed alone
    arena = getattr(p, "_osp_arena", (None,))[0]               # (FlatArena, offset) set by optim.FlatArena
    for rk2, (ref, k2) in _PACK_REG.items():
        q = ref()
        if q is None:
            dead.append(rk2)
            continue
        if q is not p and (arena is None or getattr(q, "_osp_arena", (None,))[0] is not arena):
            continue
        st = (epoch, q._version, q.data_ptr())
        c2 = getattr(q, "_osp_packs", None)
        if c2 is None or c2[0] != st:
            c2 = (st, {})
            q._osp_packs = c2
        if k2 in c2[1] and not force:                            # (force: every registered pack of the arena, fresh or not)
            continue
        _region_mark(q, ("view",) + k2)
        off, n2, t2, c_in, strd = k2[:5]
        as_f32 = len(k2) > 5
        out = _persistent(q, "_osp_pack_bufs", k2, (n2, t2, c_in), torch.float32 if as_f32 else torch.bfloat16)
        rows.append([q.data_ptr() + off, 0, out.data_ptr(), n2, t2, c_in, strd[0], strd[1], strd[2], int(as_f32)])
        fills.append((c2[1], k2, out))
        _keep(q, out)
    for rk2 in dead:
        del _PACK_REG[rk2]
    d = np.asarray(rows, dtype=np.int64)
    call("osp_pack_bf16_multi", d, len(rows))
    for dct, k2, out in fills:
        dct[k2] = out
    return p._osp_packs[1][key]


_WGRAD_BF16_MIN_M = int(__import__('os').environ.get('OSP_WGRAD_BF16_MIN_M', '2048'))
_WGRAD_CAST = __import__('os').environ.get('OSP_WGRAD_CAST', '1') != '0'


def _dense_rows(t):
    return t.dim() == 2 and t.stride(-1) == 1 and t.stride(-2) == t.shape[-1]


def _bf16_rows(t, rowscale=None):
    """(M, C) operand of a weight-gradient GEMM as bf16 (optionally row-scaled first)."""
    if t.dtype == torch.bfloat16:
        assert rowscale is None
        return t
    if rowscale is not None:
        return cast_bf16_rows(t, rowscale)
    y = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
    call("osp_cast_bf16", t, y, t.numel())
    return y


def conv_wgrad(dy, x, dw, db=None, *, T=None, taps=1, pad=0, arow=None, oscale=None, batch=1):
    """dw[n, j, c] += oscale[n] * sum_m arow[m] dy[m, n] x[m + j - pad, c];  db[n] += oscale[n] * sum_m arow[m] dy[m, n]."""
    _f32(dy, x, dw, db, arow, oscale)
    M, N = dy.shape[-2], dy.shape[-1]
    cin = x.shape[-1]
    T = M if T is None else T
    assert dw.is_contiguous() and dw.numel() == batch * N * taps * cin, (dw.shape, N, taps, cin)
    sy = dy.stride(0) if batch > 1 else 0
    sx = x.stride(0) if batch > 1 else 0
    if _precision.is_bf16() and M >= _WGRAD_BF16_MIN_M and N >= 64 and cin >= 64:
        if _WGRAD_CAST and batch == 1 and N % 64 == 0 and cin % 64 == 0 and _dense_rows(dy) and _dense_rows(x):
            # f32 activations: one bf16 copy of each operand (the row factor folded into dy's), then the ring kernel -- the register-staged
            # f32 loader of the tile-per-tap kernel rounds to bf16 at the same place (after the row factor), so the products are
            # the same; 34 -> ~20 us per launch at the predictor shapes (tools/probes/wgrad_calls.py (git history))
            conv_wgrad_bf16(_bf16_rows(dy, arow), _bf16_rows(x), dw, db, M=M, Trows=T, Tin=T, n=N, cin=cin, taps=taps, pad=pad, oscale=oscale)
            return
        call("osp_conv_wgrad_bf16", dy, 0, dy.stride(-2), x, 0, x.stride(-2), M, T, T, N, cin, taps, pad, 1, arow, oscale, dw,
             taps * cin, db, batch, sy, sx, N * taps * cin if batch > 1 else 0, N if batch > 1 else 0)
        return
    # exact f32: the ring kernel with a split workspace where it applies (csrc/wgrad_ring.hip: no atomics), else the tile-per-tap kernel
    ws = wgrad_workspace(N, taps, cin, batch, dy.device)
    call("osp_conv_wgrad_f32_split_ws" if _precision.f32_split("wgrad") else "osp_conv_wgrad_f32_ws", dy, dy.stride(-2), x, x.stride(-2), M, T, N, cin, taps, pad, arow, oscale, dw,
         taps * cin, db, batch, sy, sx, N * taps * cin if batch > 1 else 0, N if batch > 1 else 0, ws, ws.numel())


def dwconv7_ln_fwd(x, dw, dwb, lnw, lnb, eps, save, h_bf16=False):
    _f32(x, dw, dwb, lnw, lnb)
    B, T, C = x.shape
    assert x.is_contiguous() and dw.shape == (7, C)
    h = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16 if h_bf16 else torch.float32)
    xhat = torch.empty_like(x) if save else None
    rstd = torch.empty((B, T), device=x.device, dtype=torch.float32) if save else None
    call("osp_dwconv7_ln_fwd", x, dw, dwb, lnw, lnb, float(eps), h, int(h_bf16), xhat, rstd, B, T, C)
    return h, xhat, rstd


def dwconv7_bwd(dc, x, dw, dres, dres_rowmask, ddw, ddb):
    _f32(dc, x, dw, dres, dres_rowmask, ddw, ddb)
    B, T, C = x.shape
    dx = torch.empty_like(x)
    call("osp_dwconv7_bwd", dc, x, dw, dres, dres_rowmask, dx, ddw, ddb, B, T, C)
    return dx


def ln_dwconv7_bwd(dh, xhat, rstd, lnw, x, dw, dres, dres_rowmask, dlnw, dlnb, ddw, ddb):
    """LayerNorm backward + depthwise-conv backward of a ConvNeXt block in one pass (C <= 384); dlnw/dlnb/ddw/ddb accumulate."""
    _f32(dh, xhat, rstd, lnw, x, dw, dres, dres_rowmask, dlnw, dlnb, ddw, ddb)
    B, T, C = x.shape
    assert dh.is_contiguous() and xhat.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    nb = 256 if (_LNDW_TWO_STAGE and (dlnw is not None or ddw is not None) and B * T >= 4096) else 0
    ws = torch.empty((nb * 10 * C,), device=x.device, dtype=torch.float32) if nb else None
    call("osp_ln_dwconv7_bwd", dh, xhat, rstd, lnw, x, dw, dres, dres_rowmask, dx, dlnw, dlnb, ddw, ddb, B, T, C, ws, nb)
    return dx


def _lstm_chunks(B, H):
    """Utterances per launch: whole teams resident (B * H/32 <= 256 workgroups), whole XCD rounds (multiple of 8)."""
    per = max(8, (256 // (H // 32)) // 8 * 8)
    return [(lo, min(lo + per, B)) for lo in range(0, B, per)]


def _pad8(t):
    """Pad the leading (utterance) dimension to a multiple of 8 with zeros (the teams of one XCD round)."""
    B = t.shape[0]
    if B % 8 == 0:
        return t
    return torch.cat([t, t.new_zeros((8 - B % 8,) + tuple(t.shape[1:]))], 0)


def lstm_fwd(gx, whh, save=True):
    """LSTM recurrence over gx (B, T, 4H) = x W_ih^T + b_ih + b_hh with W_hh (4H, H), zero initial state.
    -> hs (B, T, H), gates (B, T, 4H) post-activation (i, f, g, o), cs (B, T, H)   (gates / cs None unless ``save``)."""
    _f32(gx, whh)
    B, T, G4 = gx.shape
    H = G4 // 4
    assert whh.shape == (G4, H) and whh.is_contiguous() and gx.is_contiguous()
    hs = torch.empty((B, T, H), device=gx.device, dtype=torch.float32)
    gates = torch.empty((B, T, G4), device=gx.device, dtype=torch.float32) if save else None
    cs = torch.empty((B, T, H), device=gx.device, dtype=torch.float32) if save else None
    for lo, hi in _lstm_chunks(B, H):
        n = hi - lo
        g = _pad8(gx[lo:hi])
        nb = g.shape[0]
        padded = nb != n
        h_ = torch.empty((nb, T, H), device=gx.device, dtype=torch.float32) if padded else hs[lo:hi]
        ga = (torch.empty((nb, T, G4), device=gx.device, dtype=torch.float32) if padded else gates[lo:hi]) if save else None
        c_ = (torch.empty((nb, T, H), device=gx.device, dtype=torch.float32) if padded else cs[lo:hi]) if save else None
        hx = torch.empty((nb, 2, H), device=gx.device, dtype=torch.float32)
        flags = torch.zeros((nb, H // 32), device=gx.device, dtype=torch.int32)
        call("osp_lstm_fwd", g, whh, h_, ga, c_, hx, flags, nb, T, H)
        if padded:
            hs[lo:hi] = h_[:n]
            if save:
                gates[lo:hi] = ga[:n]
                cs[lo:hi] = c_[:n]
    return hs, gates, cs


def lstm_bwd(dhs, gates, cs, whh):
    """Gradient w.r.t. the gate pre-activations of every step: dg (B, T, 4H)."""
    _f32(dhs, gates, cs, whh)
    B, T, H = dhs.shape
    assert dhs.is_contiguous() and gates.is_contiguous() and cs.is_contiguous()
    dg = torch.empty((B, T, 4 * H), device=dhs.device, dtype=torch.float32)
    for lo, hi in _lstm_chunks(B, H):
        n = hi - lo
        d_, g_, c_ = _pad8(dhs[lo:hi]), _pad8(gates[lo:hi]), _pad8(cs[lo:hi])
        nb = d_.shape[0]
        out = torch.empty((nb, T, 4 * H), device=dhs.device, dtype=torch.float32) if nb != n else dg[lo:hi]
        flags = torch.zeros((nb, H // 32), device=dhs.device, dtype=torch.int32)
        call("osp_lstm_bwd", d_, g_, c_, whh, out, flags, nb, T, H)
        if nb != n:
            dg[lo:hi] = out[:n]
    return dg


def dwconv_fwd(x, w, bias=None, rowmask=None, flip=False):
    """Depthwise conv of odd width K on channels-last frames: x (B,T,C), w (K,C) tap-major; flip=True is the input gradient
    (x = dy, the row mask then multiplies the input rows).  One read and one write of the activations."""
    _f32(x, w, bias, rowmask)
    B, T, C = x.shape
    assert x.is_contiguous() and w.is_contiguous() and w.shape[1] == C
    y = torch.empty_like(x)
    call("osp_dwconv_fwd", x, w, bias, rowmask, y, B, T, C, w.shape[0], int(bool(flip)))
    return y


def dwconv_wgrad(dy, x, dw, db=None, rowmask=None):
    """dw (K,C) += sum_{b,t} dy * rowmask * x[t + j - K/2];  db += sum dy * rowmask."""
    _f32(dy, x, dw, db, rowmask)
    B, T, C = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous()
    call("osp_dwconv_wgrad", dy, x, rowmask, dw, db, B, T, C, dw.shape[0])


def dropout_add(x, p, seed, stream_id, res=None):
    """res + x * keep (keep = 0 w.p. p else 1/(1-p), Philox key (seed, stream_id), counter = element index / 4)."""
    _f32(x, res)
    assert x.is_contiguous() and x.numel() % 4 == 0 and (res is None or res.is_contiguous())
    y = torch.empty_like(x)
    hs, ds = _seed(seed)
    call("osp_dropout_add", x, res, y, x.numel(), float(p), hs, ds, int(stream_id))
    return y


def layernorm_fwd(x, w, b, eps, *, save=True, rowmask=None, drop_p=0.0, seed=0, stream_id=0):
    _f32(x, w, b, rowmask)
    C = x.shape[-1]
    rows = x.numel() // C
    assert x.is_contiguous()
    y = torch.empty_like(x)
    mean = torch.empty((rows,), device=x.device, dtype=torch.float32) if save else None
    rstd = torch.empty((rows,), device=x.device, dtype=torch.float32) if save else None
    sh, sd = _seed(seed)
    call("osp_layernorm_fwd", x, w, b, float(eps), y, mean, rstd, rowmask, float(drop_p), sh, sd, int(stream_id),
         rows, C)
    return y, mean, rstd


#: layernorm_bwd at 32 x 800 x 256 (profiles/r03_lndw_variants.txt): atomics from 256 persistent workgroups 25.1 us; two-stage with
#: 512 / 1024 / 2048 workgroups 20.4 / 21.1 / 20.8 us.  The fused LN + dwconv backward does not gain (42 vs 41 us): atomics stay.
_LNBWD_BLOCKS = int(__import__('os').environ.get('OSP_LNBWD_BLOCKS', '512'))
_LNDW_TWO_STAGE = __import__('os').environ.get('OSP_LNDW_TWO_STAGE', '0') == '1'


def layernorm_bwd(dy, xin, mean, rstd, w, dlnw, dlnb, *, relu_src=None, rowmask=None, drop_p=0.0, seed=0,
                  stream_id=0):
    _f32(dy, xin, mean, rstd, w, relu_src, rowmask, dlnw, dlnb)
    C = dy.shape[-1]
    rows = dy.numel() // C
    assert dy.is_contiguous() and xin.is_contiguous()
    dx = torch.empty_like(dy)
    sh, sd = _seed(seed)
    # two-stage parameter-gradient reduction (csrc/convnext.hip): per-workgroup partial rows + a small reduce kernel instead of
    # device-scope atomics from every workgroup, which also lifts the one-workgroup-per-CU cap of the grid
    nb = min(_LNBWD_BLOCKS, (rows + 31) // 32) if (dlnw is not None and rows >= 4096) else 0
    ws = torch.empty((nb * 2 * C,), device=dy.device, dtype=torch.float32) if nb else None
    call("osp_layernorm_bwd", dy, xin, mean, rstd, w, relu_src, rowmask, float(drop_p), sh, sd, int(stream_id),
         dx, dlnw, dlnb, rows, C, ws, nb)
    return dx


# ------------------------------------------------------------------------------------------------ alignment
_LGAMMA = {}


def lgamma_table(device, n):
    """f64 table lg[i] = lgamma(i) on the device (cached; grown on demand)."""
    key = str(device)
    t = _LGAMMA.get(key)
    if t is None or t.numel() < n:
        n = max(n, 8192)
        t = torch.empty((n,), device=device, dtype=torch.float64)
        call("osp_lgamma_table", t, n)
        _LGAMMA[key] = t
    return t


def _lens(*ts):
    for t in ts:
        assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous()


def betabinom_prior(x_len, y_len, Tm, Nm):
    _lens(x_len, y_len)
    B = x_len.numel()
    lg = lgamma_table(x_len.device, Tm + Nm + 4)
    out = torch.empty((B, Tm, Nm), device=x_len.device, dtype=torch.float32)
    call("osp_betabinom_prior", lg, lg.numel(), x_len, y_len, out, B, Tm, Nm)
    return out


def pairwise_score(f, e, x_len):
    _f32(f, e)
    _lens(x_len)
    B, Tm, C = f.shape
    Nm = e.shape[1]
    assert f.is_contiguous() and e.is_contiguous()
    score = torch.empty((B, Tm, Nm), device=f.device, dtype=torch.float32)
    call("osp_pairwise_score", f, e, x_len, score, B, Tm, Nm, C)
    return score


def logsoftmax_prior_fwd(score, prior):
    B, Tm, Nm = score.shape
    lp = torch.empty_like(score)
    lse = torch.empty((B, Tm), device=score.device, dtype=torch.float32)
    call("osp_logsoftmax_prior_fwd", score, prior, lp, lse, B * Tm, Nm)
    return lp, lse


def logsoftmax_prior_bwd(dlp, score, lse, x_len, y_len):
    B, Tm, Nm = score.shape
    w = torch.empty_like(score)
    wrow = torch.empty((B, Tm), device=score.device, dtype=torch.float32)
    call("osp_logsoftmax_prior_bwd", dlp.contiguous(), score, lse, x_len, y_len, w, wrow, B, Tm, Nm)
    return w, wrow


def mas(lp, x_len, y_len):
    """-> path int32 (B,Tm) [valid t < y_len], durations f32 (B,Nm), bin_item f32 (B,) = -mean_t lp[t, path[t]]."""
    _f32(lp)
    _lens(x_len, y_len)
    B, Tm, Nm = lp.shape
    assert lp.is_contiguous()
    path = torch.zeros((B, Tm), device=lp.device, dtype=torch.int32)
    dur = torch.empty((B, Nm), device=lp.device, dtype=torch.float32)
    bin_item = torch.zeros((B,), device=lp.device, dtype=torch.float32)
    from ._lib import lib
    import ctypes
    f = lib().cdll.osp_mas_workspace_bytes
    f.restype = ctypes.c_int64
    nbytes = f(ctypes.c_int64(B), ctypes.c_int64(Tm), ctypes.c_int64(Nm))
    ws = torch.empty((max(nbytes, 8) // 8,), device=lp.device, dtype=torch.int64) if nbytes else None
    call("osp_mas", lp, x_len, y_len, path, dur, bin_item, ws, B, Tm, Nm)
    return path, dur, bin_item


def bin_loss_bwd(path, y_len, gscale, dlp):
    B, Tm, Nm = dlp.shape
    call("osp_bin_loss_bwd", path, y_len, gscale, dlp, B, Tm, Nm)


def duration_stats(ds, xs0=None, xs1=None, x_len=None, y_len=None, want_centre=False):
    """token-level averages of xs0/xs1 by duration, and/or gaussian-upsampling centres cumsum(d) - d/2."""
    _f32(ds, xs0, xs1)
    B, Nm = ds.shape
    Tm = xs0.shape[1] if xs0 is not None else 0
    a0 = torch.empty((B, Nm), device=ds.device, dtype=torch.float32) if xs0 is not None else None
    a1 = torch.empty((B, Nm), device=ds.device, dtype=torch.float32) if xs1 is not None else None
    ce = torch.empty((B, Nm), device=ds.device, dtype=torch.float32) if want_centre else None
    call("osp_duration_stats", ds.contiguous(), xs0, xs1, x_len, y_len, a0, a1, ce, B, Tm, Nm)
    return a0, a1, ce


def gaussian_weights(centre, x_len, y_len, Tm, delta=0.1):
    B, Nm = centre.shape
    P = torch.empty((B, Tm, Nm), device=centre.device, dtype=torch.float32)
    call("osp_gaussian_weights", centre, x_len, y_len, float(delta), P, B, Tm, Nm)
    return P


def gather_rows(src, start, S, mult=1):
    """out[b, s, :] = src[b, start[b]*mult + s, :]; src (B,T,C)."""
    _f32(src)
    _lens(start)
    B, T, C = src.shape
    assert src.is_contiguous()
    out = torch.empty((B, S, C), device=src.device, dtype=torch.float32)
    call("osp_gather_rows", src, start, mult, out, B, T, S, C)
    return out


def drop_path_rows(drop_p, rowmask, B, T, seed, stream_id, device):
    """DropPath factors of L blocks in one launch: scale (L, B*T) = 0 w.p. drop_p[l] else 1/(1-drop_p[l]) per (block, utterance),
    and rowf = scale * rowmask (None without a mask).  Philox key (seed, stream_id): graph-replay safe (device seed)."""
    L = len(drop_p)
    scale = torch.empty((L, B * T), device=device, dtype=torch.float32)
    rowf = torch.empty((L, B * T), device=device, dtype=torch.float32) if rowmask is not None else None
    if rowmask is not None:
        _f32(rowmask)
    hs, ds = _seed(seed)
    p = _host_f32(drop_p)
    call("osp_drop_path_rows", p, rowmask, L, B, T, hs, ds, int(stream_id), scale, rowf)
    return scale, rowf


def period_fold(x, period, backward=False, T=None):
    """forward: wave (B, T) -> period-column sequences (B * period, ceil(T / period)) with the reference's right reflect pad;
    backward: gradient of those sequences -> (B, T)."""
    _f32(x)
    assert x.is_contiguous()
    if not backward:
        B, T = x.shape
        y = torch.empty((B * period, -(-T // period)), device=x.device, dtype=torch.float32)
    else:
        B = x.shape[0] // period
        y = torch.empty((B, T), device=x.device, dtype=torch.float32)
    call("osp_period_fold", x, y, B, T, period, int(backward))
    return y


def segment_starts(r01, lengths, segment_size, lead=4):
    """long(r01 * clamp(float(len - lead) - segment_size, 0)) per utterance (utils/segments.py:29-34), one launch."""
    _f32(r01)
    _lens(lengths)
    out = torch.empty_like(lengths)
    call("osp_segment_starts", r01.contiguous(), lengths, lengths.numel(), int(lead), int(segment_size), out)
    return out


def expand_by_duration(x, dur, Tout):
    _f32(x)
    _lens(dur)
    B, Nm, C = x.shape
    out = torch.empty((B, Tout, C), device=x.device, dtype=torch.float32)
    call("osp_expand_by_duration", x.contiguous(), dur, out, B, Nm, Tout, C)
    return out


# ------------------------------------------------------------------------------------------------ losses
def variance_losses(d_hat, p_hat, e_hat, ds, ps, es, x_len, clip_val=1e-8):
    _f32(d_hat, p_hat, e_hat, ds, ps, es)
    _lens(x_len)
    B, T = d_hat.shape
    out = torch.empty((3,), device=d_hat.device, dtype=torch.float32)
    gd, gp, ge = torch.empty_like(d_hat), torch.empty_like(d_hat), torch.empty_like(d_hat)
    call("osp_variance_losses", d_hat.contiguous(), p_hat.contiguous(), e_hat.contiguous(), ds.contiguous(),
         ps.contiguous(), es.contiguous(), x_len, float(clip_val), out, gd, gp, ge, B, T)
    return out, gd, gp, ge


def forwardsum_ctc(lp, x_len, y_len, want_grad=True, blank_logprob=-1.0):
    """-> loss_item (B,) [ctc_b / N_b], grad (B,Tm,Nm) of sum_b loss_item/B w.r.t. lp (or None)."""
    _f32(lp)
    _lens(x_len, y_len)
    B, Tm, Nm = lp.shape
    from ._lib import lib
    import ctypes
    f = lib().cdll.osp_forwardsum_ctc_workspace_floats
    f.restype = ctypes.c_int64
    n = f(ctypes.c_int64(B), ctypes.c_int64(Tm), ctypes.c_int64(Nm))
    ws = torch.empty((n,), device=lp.device, dtype=torch.float32)
    loss_item = torch.empty((B,), device=lp.device, dtype=torch.float32)
    grad = torch.empty_like(lp) if want_grad else None
    call("osp_forwardsum_ctc", lp, x_len, y_len, float(blank_logprob), ws, loss_item, grad, B, Tm, Nm)
    return loss_item, grad


# ------------------------------------------------------------------------------------------------ text embedding
def text_embed_fwd(tok, E, pos, scale, drop_p=0.0, seed=0, stream_id=0):
    B, T = tok.shape
    C = E.shape[1]
    assert tok.dtype == torch.int64 and tok.is_contiguous() and pos.shape[0] >= T and pos.shape[1] == C
    out = torch.empty((B, T, C), device=E.device, dtype=torch.float32)
    sh, sd = _seed(seed)
    call("osp_text_embed_fwd", tok, E, pos, scale, float(C) ** 0.5, float(drop_p), sh, sd, int(stream_id), out, B, T, C)
    return out


def text_embed_bwd(dy, tok, pos, dE, dscale, padding_idx=0, drop_p=0.0, seed=0, stream_id=0):
    B, T, C = dy.shape
    sh, sd = _seed(seed)
    call("osp_text_embed_bwd", dy.contiguous(), tok, pos, float(C) ** 0.5, float(drop_p), sh, sd, int(stream_id),
         int(padding_idx), dE, dscale, B, T, C)


# ------------------------------------------------------------------------------------------------ bf16 MFMA GEMMs
EPI_LRELU, EPI_LRELU_BWD = 8, 9


def _isbf(t):
    return int(t is not None and t.dtype == torch.bfloat16)


def pack_bf16(w, n_out, taps, cin, w_strides, kscale=None):
    """(n_out, taps, cin) k-contiguous bf16 copy of a weight addressed through element strides (osp_pack_bf16); ``kscale``
    (cin floats) multiplies along the reduction index first."""
    wp = torch.empty((n_out, taps, cin), device=w.device, dtype=torch.bfloat16)
    call("osp_pack_bf16", w, kscale, wp, n_out, taps, cin, w_strides[0], w_strides[1], w_strides[2])
    return wp


def colsum_prod(a, b, rowf, out):
    """out[c] += sum_m rowf[m] * a[m, c] * b[m, c]  (f32, (M, C) contiguous)."""
    M, C = a.shape
    call("osp_colsum_prod", a, b, rowf, out, M, C)


def cast_bf16_rows(x, rowscale):
    """bf16(rowscale[m] * x[m, :]) for an (M, C) f32 matrix."""
    M, C = x.shape
    y = torch.empty((M, C), device=x.device, dtype=torch.bfloat16)
    call("osp_cast_bf16_rows", x, rowscale, y, M, C)
    return y


def _host_i64(vals):
    import numpy as np
    return np.asarray(vals, dtype=np.int64)


def _host_f32(vals):
    import numpy as np
    return np.asarray(vals, dtype=np.float32)


def l1_sum_multi(targets, ys, out):
    """out += sum_i mean |target_i - y_i| over a list of same-dtype tensor pairs, one launch per 32 pairs."""
    a, b = _host_i64([t.data_ptr() for t in targets]), _host_i64([t.data_ptr() for t in ys])
    n, sc = _host_i64([t.numel() for t in ys]), _host_f32([1.0 / t.numel() for t in ys])
    bf = _host_i64([_isbf(t) for t in ys])
    _keep(*targets, *ys)
    call("osp_l1_sum_multi", a, b, n, sc, bf, len(ys), out)


def l1_sign_multi(targets, ys, gscale):
    """[gscale * sign(y_i - target_i) / numel_i] for the same list (feature-matching gradient), one launch per 32 pairs."""
    gbs = [torch.empty_like(t) for t in ys]
    a, b = _host_i64([t.data_ptr() for t in targets]), _host_i64([t.data_ptr() for t in ys])
    o, n = _host_i64([t.data_ptr() for t in gbs]), _host_i64([t.numel() for t in ys])
    sc = _host_f32([1.0 / t.numel() for t in ys])
    bf = _host_i64([_isbf(t) for t in ys])
    _keep(*targets, *ys, *gbs)
    call("osp_l1_sign_multi", a, b, o, n, sc, bf, len(ys), gscale)
    return gbs


def hinge_sum_multi(xs, sgns, out):
    x, n = _host_i64([t.data_ptr() for t in xs]), _host_i64([t.numel() for t in xs])
    sg, sc = _host_f32(sgns), _host_f32([1.0 / t.numel() for t in xs])
    _keep(*xs)
    call("osp_hinge_sum_multi", x, n, sg, sc, len(xs), out)


def hinge_grad_multi(xs, sgns, gscale):
    dxs = [torch.empty_like(t) for t in xs]
    x, o = _host_i64([t.data_ptr() for t in xs]), _host_i64([t.data_ptr() for t in dxs])
    n, sg, sc = _host_i64([t.numel() for t in xs]), _host_f32(sgns), _host_f32([1.0 / t.numel() for t in xs])
    _keep(*xs, *dxs)
    call("osp_hinge_grad_multi", x, o, n, sg, sc, len(xs), gscale)
    return dxs


def _wn_desc(rows):
    import numpy as np
    ptr = lambda t: 0 if t is None else t.data_ptr()                          # noqa: E731
    return np.asarray([[ptr(r[k]) for k in range(9)] + list(r[9:13]) for r in rows], dtype=np.int64)


def wnorm_fwd_multi(items, reuse=None):
    """items: [(v, g, want_f32, want_t)] -> [(wn, wn32, wt, inv)]: the weight-norm packs of many convs in one launch per 32.
    ``reuse``: per item the previous pack tuple (or None) -- its buffers are written in place when they have what is wanted, so a
    conv's packs keep their addresses over the optimizer epochs (the call tapes of the stacks hold those addresses)."""
    outs, rows = [], []
    for k, (v, g, want_f32, want_t) in enumerate(items):
        Cout, Cin, P, Q = v.shape
        dev = v.device
        old = reuse[k] if reuse is not None else None
        if old is not None and old[0] is not None and old[0].device == dev and (old[1] is not None or not want_f32) and (old[2] is not None or not want_t):
            wn, wn32, wt, inv = old
        else:
            wn = torch.empty((Cout, Q, P, Cin), device=dev, dtype=torch.bfloat16)
            wn32 = torch.empty((Cout, Q, P, Cin), device=dev, dtype=torch.float32) if want_f32 else None
            wt = torch.empty((Cin, Q, P, Cout), device=dev, dtype=torch.bfloat16) if want_t else None
            inv = torch.empty((Cout,), device=dev, dtype=torch.float32)
        outs.append((wn, wn32, wt, inv))
        rows.append((v, g, wn, wn32, wt, inv, None, None, None, Cout, Cin, P, Q))
    d = _wn_desc(rows)
    _keep(*[t for r in rows for t in r[:9]])
    call("osp_wnorm_fwd_multi", d, len(rows))
    return outs


def wnorm_bwd_multi(items):
    """items: [(dwn, v, g, inv, dv, dg)] (dv / dg accumulated), one launch per 32 convs."""
    rows = []
    for dwn, v, g, inv, dv, dg in items:
        Cout, Cin, P, Q = v.shape
        rows.append((v, g, None, None, None, inv, dwn, dv, dg, Cout, Cin, P, Q))
    d = _wn_desc(rows)
    _keep(*[t for r in rows for t in r[:9]])
    call("osp_wnorm_bwd_multi", d, len(rows))


def param_bf16(p, transposed=False):
    """bf16 copy of a 2-D f32 parameter (N, K) -- or of its transpose -- cached on the Parameter object for the current
    optimizer epoch (same invalidation rule as disc_ops.wnorm_packed)."""
    from . import values
    stamp = (values.param_epoch(), p._version, p.data_ptr())
    cache = getattr(p, "_osp_bf16", None)
    if cache is None or cache[0] != stamp:
        cache = (stamp, {})
        p._osp_bf16 = cache
    if transposed not in cache[1] or _tape_recording():      # (recording: param_bf16_many decides, once per region and pack)
        param_bf16_many([(p, transposed, None)])
    return p._osp_bf16[1][transposed]


def param_bf16_many(requests):
    """Fill the per-parameter bf16 pack cache (param_bf16 / the gamma-scaled transposed pack of the block backward) for a list of
    requests in ONE launch: requests = [(param (N, K), transposed: bool, kscale tensor or None)].  Entries that are already
    valid for the current optimiser epoch are skipped."""
    import numpy as np
    from . import values
    rows, fills = [], []
    for p, transposed, kscale in requests:
        stamp = (values.param_epoch(), p._version, p.data_ptr())
        cache = getattr(p, "_osp_bf16", None)
        if cache is None or cache[0] != stamp:
            cache = (stamp, {})
            p._osp_bf16 = cache
        key = transposed if kscale is None else ("t_scaled", kscale.data_ptr(), kscale._version)
        bkey = key if kscale is None else ("t_scaled", kscale.data_ptr())
        if key in cache[1] and not _region_first_use(p, ("bf16", bkey)):
            continue
        _region_mark(p, ("bf16", bkey))
        N, Kd = p.shape
        if transposed or kscale is not None:                     # out[k_dim = N index ... ] : (Kd rows, N reduction) = W^T
            out = _persistent(p, "_osp_bf16_bufs", bkey, (Kd, N), torch.bfloat16)
            rows.append([p.data_ptr(), 0 if kscale is None else kscale.data_ptr(), out.data_ptr(), Kd, 1, N, 1, 0, Kd, 0])
        else:
            out = _persistent(p, "_osp_bf16_bufs", bkey, (N, Kd), torch.bfloat16)
            rows.append([p.data_ptr(), 0, out.data_ptr(), N, 1, Kd, Kd, 0, 1, 0])
        fills.append((cache[1], key, out))
        _keep(p, kscale, out)
    if not rows:
        return
    d = np.asarray(rows, dtype=np.int64)
    call("osp_pack_bf16_multi", d, len(rows))
    for dct, key, out in fills:
        dct[key] = out


def param_bf16_scaled_t(p, kscale):
    """gamma-scaled transposed pack of a (C, I) weight: out[i, c] = p[c, i] * kscale[c] (cached per optimiser epoch)."""
    from . import values
    stamp = (values.param_epoch(), p._version, p.data_ptr())
    cache = getattr(p, "_osp_bf16", None)
    key = ("t_scaled", kscale.data_ptr(), kscale._version)
    if cache is not None and cache[0] == stamp and key in cache[1] and not _tape_recording():
        return cache[1][key]
    param_bf16_many([(p, True, kscale)])
    return p._osp_bf16[1][key]


def cast_bf16(x):
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    call("osp_cast_bf16", x.contiguous(), y, x.numel())
    return y


def conv_gemm_bf16(a, w, n_out, *, M, Trows, Tin, cin, taps=1, a_step=1, a_tapstep=1, a_off=0, lda=None,
                   w_strides=None, out=None, out_bf16=False, ldc=None, Tc=None, c_step=1, c_off=0, out_rows=None,
                   epi=EPI_NONE, bias=None, gamma=None, res=None, rowmask=None, rowscale=None, aux_out=None, aux_in=None,
                   a_rowscale=None, slope=0.1, accumulate=False, batch=1, batch_strides=(0, 0, 0, 0)):
    """bf16-MFMA version of conv_gemm with conv stride / output row mapping (see csrc/gemm_bf16.hip)."""
    lda = a.stride(-2) if lda is None else lda
    if w_strides is None:
        w_strides = (taps * cin, cin, 1)
    Tc = Trows if Tc is None else Tc
    if out is None:
        rows = M if out_rows is None else out_rows
        out = torch.empty((rows, n_out), device=a.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    ldc = out.stride(-2) if ldc is None else ldc
    ld_aux = 0
    for t in (aux_out, aux_in):
        if t is not None:
            ld_aux = t.stride(-2)
    ldr = res.stride(-2) if res is not None else 0
    call("osp_conv_gemm_bf16", a, _isbf(a), lda, M, Trows, Tin, cin, taps, a_step, a_tapstep, a_off, a_rowscale, w,
         _isbf(w), w_strides[0], w_strides[1], w_strides[2], n_out, out, _isbf(out), ldc, Tc, c_step, c_off, epi, bias,
         gamma, res, _isbf(res), ldr, rowmask, rowscale, aux_out, aux_in, _isbf(aux_in if aux_in is not None else aux_out),
         ld_aux, float(slope), batch,
         batch_strides[0], batch_strides[1], batch_strides[2], batch_strides[3], bool(accumulate))
    return out


_WGRAD_WS_CAP = 32 << 20
_WGRAD_WS = {}          # (device index, launch-stream key) -> persistent uint8 scratch, grown to the largest request (<= _WGRAD_WS_CAP)


def wgrad_workspace(n, taps, cin, batch, device):
    """Scratch for the frame splits of one weight-gradient launch (csrc/wgrad_ring.hip): up to 64 partial copies of dW (+ db),
    32 MB at most; the library takes as many splits as fit.  ONE persistent buffer per LAUNCH stream: the partial-sum
    kernel and its reduction are queued back to back on that stream, so stream order is all the protection the scratch needs, and
    nothing is allocated per launch -- a deferred launch (ops.side_wgrad) or a call tape holds a pointer into a buffer that is never
    freed instead of pinning 32 MB each.  The launch stream is the side stream of the calling stream while ops.side_wgrad is
    recording (ops._flush_wgrad sends a calling stream's launches to exactly one side stream), else the stream the call goes to."""
    from . import _lib
    blk = (n * taps * cin + n + 3) // 4 * 16 * batch
    need = min(_WGRAD_WS_CAP, 64 * blk)
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing():
        # a buffer made during a capture would live in the graph's private pool: not cached (one per launch, owned by the graph, as before)
        return torch.empty((need,), device=device, dtype=torch.uint8)
    raw = _lib._STREAM_OVERRIDE[0] or _lib._raw_stream(idx)
    key = (idx, ("side", raw) if _lib._RECORD[0] is not None else raw)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        # grown, never shrunk; the old buffer may still be in use by launches queued (or taped) earlier: kept alive, not freed
        if ws is not None:
            _WGRAD_WS.setdefault("retired", []).append(ws)
        ws = _WGRAD_WS[key] = torch.empty((_WGRAD_WS_CAP if ws is not None else need,), device=device, dtype=torch.uint8)
    return ws[:need]


def conv_wgrad_bf16(dy, x, dw, db=None, *, M, Trows, Tin, n, cin, taps=1, pad=0, x_step=1, arow=None, oscale=None,
                    batch=1, strides=(0, 0, 0, 0)):
    if _WGRAD_CAST and batch == 1 and _isbf(dy) != _isbf(x) and n % 64 == 0 and cin % 64 == 0 and _dense_rows(dy) and _dense_rows(x):
        # one bf16 and one f32 operand (a ConvNeXt block whose forward ran on the exact-f32 index path): the mixed pair took the
        # generic tile kernel (67 us per launch at 4096 x 256 x 1024, eight per step); a bf16 copy of the f32 side puts it on the ring kernel
        dy, x, arow = _bf16_rows(dy, arow if not _isbf(dy) else None), _bf16_rows(x), (arow if _isbf(dy) else None)
    if _isbf(dy) and _isbf(x) and arow is None:
        ws = wgrad_workspace(n, taps, cin, batch, dy.device)
        call("osp_conv_wgrad_bf16_ws", dy, 1, dy.stride(-2), x, 1, x.stride(-2), M, Trows, Tin, n, cin, taps,
             pad, x_step, arow, oscale, dw, taps * cin, db, batch, strides[0], strides[1], strides[2], strides[3], ws, ws.numel())
        return
    call("osp_conv_wgrad_bf16", dy, _isbf(dy), dy.stride(-2), x, _isbf(x), x.stride(-2), M, Trows, Tin, n, cin, taps,
         pad, x_step, arow, oscale, dw, taps * cin, db, batch, strides[0], strides[1], strides[2], strides[3])


def conv2d_gemm_bf16(a, w, n_out, *, M, Trows, Wrows, Hin, Win, cin, taps, KW, a_step_h, a_tapstep_h, a_off_h, a_step,
                     a_tapstep, a_off, w_strides, out, ldc, Tc, Wc, c_step_h=1, c_off_h=0, c_step=1, c_off=0,
                     epi=EPI_NONE, bias=None, res=None, aux_in=None, slope=0.1, lda=None):
    """2-D (channels-last conv2d) form of conv_gemm_bf16; w_strides = (sBn, sBtap_h, sBtap_w, sBk)."""
    lda = a.stride(-2) if lda is None else lda
    ld_aux = aux_in.stride(-2) if aux_in is not None else 0
    ldr = res.stride(-2) if res is not None else 0
    call("osp_conv2d_gemm_bf16", a, _isbf(a), lda, M, Trows, Wrows, Hin, Win, cin, taps, KW, a_step_h, a_tapstep_h, a_off_h,
         a_step, a_tapstep, a_off, w, _isbf(w), w_strides[0], w_strides[1], w_strides[2], w_strides[3], n_out, out,
         _isbf(out), ldc, Tc, Wc, c_step_h, c_off_h, c_step, c_off, epi, bias, res, _isbf(res), ldr, aux_in, _isbf(aux_in),
         ld_aux, float(slope))
    return out


def conv2d_wgrad_bf16(dy, x, dw, db, *, M, Trows, Wrows, Hin, Win, n, cin, taps, KW, pad_h, pad_w, step_h, step_w):
    if _isbf(dy) and _isbf(x):
        ws = wgrad_workspace(n, taps, cin, 1, dy.device)
        call("osp_conv2d_wgrad_bf16_ws", dy, 1, dy.stride(-2), x, 1, x.stride(-2), M, Trows, Wrows, Hin, Win, n, cin,
             taps, KW, pad_h, pad_w, step_h, step_w, dw, db, ws, ws.numel())
        return
    call("osp_conv2d_wgrad_bf16", dy, _isbf(dy), dy.stride(-2), x, _isbf(x), x.stride(-2), M, Trows, Wrows, Hin, Win, n, cin,
         taps, KW, pad_h, pad_w, step_h, step_w, dw, db)


def smallcin_fwd(x, w, b, *, U, Hin, Win, Ho, Wo, cout, KH, KW, sh, sw, ph, pw, slope, out_bf16):
    """Cin = 1 direct conv: x (U,Hin,Win) f32, w (cout, KH*KW) f32 -> (U*Ho*Wo, cout)."""
    y = torch.empty((U * Ho * Wo, cout), device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    call("osp_smallcin_conv_fwd", x, w, b, y, _isbf(y), U * Ho * Wo, Ho * Wo, Wo, Hin, Win, cout, KH * KW, KW, sh, sw, ph, pw,
         slope is not None, float(slope or 0.0))
    return y


_SMALLCIN_WGS = int(__import__('os').environ.get('OSP_SMALLCIN_WGS', '512'))      # workgroup cap of the MFMA weight-gradient kernel (csrc/smallcin.hip)


def smallcin_wgrad(x, dy, dw, db, *, U, Hin, Win, Ho, Wo, cout, KH, KW, sh, sw, ph, pw):
    ws = torch.empty((_SMALLCIN_WGS * 4096,), device=x.device, dtype=torch.float32) if _isbf(dy) else None     # two-stage reduction scratch
    call("osp_smallcin_conv_wgrad", x, dy, _isbf(dy), dw, db, U * Ho * Wo, Ho * Wo, Wo, Hin, Win, cout, KH * KW, KW, sh, sw, ph,
         pw, ws, 0 if ws is None else ws.numel())


# ------------------------------------------------------------------------------------------------ weight norm / L1
def wnorm_fwd(v, g, want_f32=False, want_t=True):
    """v (Cout,Cin,P,Q) f32, g (Cout,1,1,1) -> wn bf16 (Cout,Q,P,Cin), wn32 or None, wt bf16 (Cin,Q,P,Cout) or None, inv_norm."""
    Cout, Cin, P, Q = v.shape
    dev = v.device
    wn = torch.empty((Cout, Q, P, Cin), device=dev, dtype=torch.bfloat16)
    wn32 = torch.empty((Cout, Q, P, Cin), device=dev, dtype=torch.float32) if want_f32 else None
    wt = torch.empty((Cin, Q, P, Cout), device=dev, dtype=torch.bfloat16) if want_t else None
    inv = torch.empty((Cout,), device=dev, dtype=torch.float32)
    call("osp_wnorm_fwd", v, g, wn, wn32, wt, inv, Cout, Cin, P, Q)
    return wn, wn32, wt, inv


def wnorm_bwd(dwn, v, g, inv, dv, dg):
    Cout, Cin, P, Q = v.shape
    call("osp_wnorm_bwd", dwn, v, g, inv, dv, dg, Cout, Cin, P, Q)


def l1_sum(a, b, scale, out):
    call("osp_l1_sum", a, b, _isbf(a), a.numel(), float(scale), out)


def l1_sign(a, b, scale, gscale):
    gb = torch.empty_like(b)
    call("osp_l1_sign", a, b, _isbf(a), a.numel(), float(scale), gscale, gb)
    return gb


# ------------------------------------------------------------------------------------------------ element-wise plumbing (csrc/ew.hip)
def ew_axpby(x, y, a=1.0, b=1.0, out=None):
    """a * x + b * y (y None: a * x + b) on contiguous f32 tensors of one shape."""
    _f32(x, y)
    assert x.is_contiguous() and (y is None or (y.is_contiguous() and y.shape == x.shape))
    out = torch.empty_like(x) if out is None else out
    call("osp_ew_axpby", x, y, out, x.numel(), float(a), float(b))
    return out


def ew_mul_rows(x, rowvec, out=None):
    """x (M, C) * rowvec (M,): one factor per row."""
    _f32(x, rowvec)
    assert x.is_contiguous() and rowvec.is_contiguous() and rowvec.numel() * x.shape[-1] == x.numel()
    out = torch.empty_like(x) if out is None else out
    call("osp_ew_mul", x, rowvec, out, x.numel(), x.shape[-1], 1)
    return out


def ew_scale_dev(x, s, c=1.0, out=None):
    """c * s[0] * x with ``s`` a one-element f32 device tensor (an incoming loss gradient)."""
    _f32(x, s)
    assert x.is_contiguous() and s.numel() == 1
    out = torch.empty_like(x) if out is None else out
    call("osp_ew_scale_dev", x, s, out, x.numel(), float(c))
    return out


def relu_mask(g, y):
    """g where y > 0, else 0 (ReLU backward)."""
    _f32(g, y)
    assert g.is_contiguous() and y.is_contiguous() and g.numel() == y.numel()
    out = torch.empty_like(g)
    call("osp_ew_relu_mask", g, y, out, g.numel())
    return out


def transpose_last2(x):
    """(B, R, C) -> (B, C, R) contiguous."""
    _f32(x)
    B, R, C = x.shape
    assert x.is_contiguous()
    y = torch.empty((B, C, R), device=x.device, dtype=torch.float32)
    call("osp_transpose_last2", x, y, B, R, C)
    return y


def length_masks(lengths, T):
    """(padding mask (B, T) bool, True = pad; keep mask (B*T,) f32) of int64 lengths in one launch (utils/model.py:12-16)."""
    _lens(lengths)
    B = lengths.numel()
    keep = torch.empty((B * T,), device=lengths.device, dtype=torch.float32)
    pad = torch.empty((B, T), device=lengths.device, dtype=torch.bool)
    call("osp_length_masks", lengths, B, T, keep, pad)
    return pad, keep


def posenc_fwd(x, pe, alpha):
    """x (B, T, C) + alpha[0] * pe (T, C) over the batch (_transformer/embedding.py:120-124); alpha: device scalar."""
    _f32(x, pe, alpha)
    B, T, C = x.shape
    assert x.is_contiguous() and pe.is_contiguous() and tuple(pe.shape) == (T, C) and alpha.numel() == 1
    y = torch.empty_like(x)
    call("osp_posenc_fwd", x, pe, alpha, y, B, T * C)
    return y


def posenc_dalpha(dy, pe, nparts=128):
    """sum_{b, t, c} dy[b, t, c] * pe[t, c] as a 0-dim tensor: one partial per workgroup, then osp_sum_scaled (fixed order)."""
    _f32(dy, pe)
    B, T, C = dy.shape
    assert dy.is_contiguous() and pe.is_contiguous() and tuple(pe.shape) == (T, C)
    parts = torch.empty((nparts,), device=dy.device, dtype=torch.float32)
    call("osp_posenc_dalpha", dy, pe, B, T * C, parts, nparts)
    return sum_scaled(parts, 1.0)


def permute_0213(x):
    """(A, B, C, D) contiguous f32 -> (A, C, B, D) contiguous: x.transpose(1, 2).contiguous() in one launch of ours."""
    _f32(x)
    A, B_, C, D = x.shape
    assert x.is_contiguous() and D % 4 == 0
    y = torch.empty((A, C, B_, D), device=x.device, dtype=torch.float32)
    call("osp_permute_0213", x, y, A, B_, C, D)
    return y


def sum_scaled(x, scale):
    """scale * sum(x) as a 0-dim f32 tensor (small vectors: one workgroup, deterministic order)."""
    _f32(x)
    assert x.is_contiguous()
    out = torch.empty((), device=x.device, dtype=torch.float32)
    call("osp_sum_scaled", x, x.numel(), float(scale), out)
    return out


def dot_multi(terms, coeffs):
    """sum_i coeffs[i] * terms[i][0] over device scalars -> 0-dim tensor."""
    for t in terms:
        _f32(t)
    out = torch.empty((), device=terms[0].device, dtype=torch.float32)
    _keep(*terms)
    call("osp_dot_multi", _host_i64([t.data_ptr() for t in terms]), _host_f32(coeffs), len(terms), out, None)
    return out


def scale_vec(g, coeffs):
    """(g[0] * coeffs[i])_i as a (len(coeffs),) tensor."""
    _f32(g)
    out = torch.empty((len(coeffs),), device=g.device, dtype=torch.float32)
    call("osp_scale_vec", g, _host_f32(coeffs), len(coeffs), out)
    return out


def store_i64(dst, value):
    call("osp_store_i64", dst, int(value))


# ------------------------------------------------------------------------------------------------ attention (A19)
def attn_softmax_fwd(S, klen, B, H, T1, T2, scale, drop_p=0.0, seed=0, stream_id=0):
    """in place: S (B*H, T1, T2) scores -> probabilities over the valid keys; returns (P, Pd) with Pd = dropout(P) (or P)."""
    Pd = torch.empty_like(S) if drop_p > 0.0 else None
    sh, sd = _seed(seed)
    call("osp_attn_softmax_fwd", S, Pd, klen, B, H, T1, T2, float(scale), float(drop_p), sh, sd, int(stream_id))
    return S, (Pd if Pd is not None else S)


def param_f32_t(p, kscale=None):
    """f32 transpose (K, N) of a 2-D f32 parameter (N, K), optionally with row n of p scaled by kscale[n] first -- the k-contiguous
    operand of an exact-f32 input-gradient GEMM (ConvNeXt block backward in the f32 / mixed modes).  Cached on the Parameter for the
    current optimizer epoch like the bf16 packs and refreshed in place by the library (one osp_pack_bf16_multi row with out_f32 set:
    tape-safe; two torch ops before round 5)."""
    import numpy as np
    from . import values
    stamp = (values.param_epoch(), p._version, p.data_ptr(), None if kscale is None else (kscale.data_ptr(), kscale._version))
    key = "_osp_f32_t" if kscale is None else "_osp_f32_t_scaled"
    cache = getattr(p, key, None)
    if cache is None or cache[0] != stamp or _region_first_use(p, key):
        N, Kd = p.shape
        _region_mark(p, key)
        out = _persistent(p, "_osp_f32_t_bufs", key, (Kd, N), torch.float32)
        row = [[p.data_ptr(), 0 if kscale is None else kscale.data_ptr(), out.data_ptr(), Kd, 1, N, 1, 0, Kd, 1]]
        _keep(p, kscale, out)
        call("osp_pack_bf16_multi", np.asarray(row, dtype=np.int64), 1)
        cache = (stamp, out)
        setattr(p, key, cache)
    return cache[1]


def param_bf16_kperm16(p):
    """bf16 copy of a 2-D f32 parameter (N, K) whose K axis is permuted inside every group of 16 to [0-3, 8-11, 4-7, 12-15]: the order
    in which an MFMA 32x32 accumulator tile hands its rows to the next MFMA as an A operand (csrc/mlp_fused.hip, phase 2).  Cached on
    the Parameter for the current optimizer epoch like param_bf16."""
    from . import values
    stamp = (values.param_epoch(), p._version, p.data_ptr())
    cache = getattr(p, "_osp_bf16_kperm16", None)
    if cache is None or cache[0] != stamp or _region_first_use(p, "kperm16"):
        N, Kd = p.shape
        _region_mark(p, "kperm16")
        assert Kd % 16 == 0
        w = _persistent(p, "_osp_bf16_bufs", "kperm16", (N, Kd), torch.bfloat16)       # refreshed in place: one launch, no torch op
        call("osp_pack_bf16_kperm16", p.detach(), w, N, Kd)
        cache = (stamp, w)
        p._osp_bf16_kperm16 = cache
    return cache[1]


#: the padded batch being decoded: set by generator.synthesise around its decoder / vocoder passes (``live_rows``).  The fused MLP's
#: launches over exactly M rows with a row mask walk the rows in the order "unmasked first" (a stable partition by the mask, made at the
#: first such launch and kept for the rest of the pass: the mask of a pass does not change) and know how many are unmasked
_LIVE_ROWS = [None]


class live_rows:
    """``with live_rows(M, n):`` -- the no-grad ConvNeXt MLP launches over M masked rows inside know that n of them are unmasked and
    walk those first (csrc/mlp_fused.hip: masked row blocks cost nothing).  A hint: n only picks the workgroup mix."""

    def __init__(self, M, n):
        self.v = {"M": int(M), "n": int(n), "perm": None, "mask": None}

    def __enter__(self):
        self.old, _LIVE_ROWS[0] = _LIVE_ROWS[0], self.v
        return self

    def __exit__(self, *exc):
        _LIVE_ROWS[0] = self.old
        return False


def _row_order(hint, rowmask):
    """position -> row, unmasked rows first (int32), for the mask of the running pass (same storage = same mask)."""
    key = rowmask.data_ptr()                                      # (inside one pass: same storage = same mask; inference tensors carry no version)
    if hint["perm"] is None or hint["mask"] != key:
        perm = torch.empty((rowmask.numel(),), device=rowmask.device, dtype=torch.int32)
        call("osp_row_order", rowmask, perm, rowmask.numel())
        hint["perm"] = perm
        hint["mask"], hint["keep"] = key, rowmask                  # (keeps the mask's storage alive: the key cannot be recycled)
    return hint["perm"]


def convnext_mlp_fused(h, W1, b1, W2, b2, gamma, x, rowmask=None, rowscale=None):
    """ConvNeXt block MLP without gradients in one launch (csrc/mlp_fused.hip): y = (x + rowscale * gamma * (W2 gelu(W1 h + b1) + b2)) * rowmask.
    h (M, C) bf16 (dwconv7_ln_fwd(..., h_bf16=True)), x (M, C) f32, W1 (I, C) / W2 (C, I) f32 Parameters; C in {256, 384}, I % 128 == 0."""
    M, C = x.shape
    I = W1.shape[0]
    assert h.dtype == torch.bfloat16 and h.shape == (M, C) and h.is_contiguous() and x.is_contiguous()
    _f32(x, b1, b2, gamma)
    y = torch.empty_like(x)
    hint = _LIVE_ROWS[0]
    perm, live = None, -1
    if hint is not None and hint["M"] == M and rowmask is not None and rowmask.numel() == M and os.environ.get("OSP_MLP_LIVE", "1") != "0":
        perm, live = _row_order(hint, rowmask), hint["n"]
    call("osp_convnext_mlp_fused_live", h, param_bf16(W1), b1, param_bf16_kperm16(W2), b2, gamma, x, rowmask, rowscale, y, M, C, I, perm, live)
    return y


def mlp_fused_supported(C, I):
    return C in (256, 384) and I % 128 == 0 and 128 <= I <= 4096


def attn_fused_fwd(q, k, v, klen, H):
    """softmax(q k^T / sqrt(dk) over the valid keys) v per head, scores never written: q, k, v (B, T, H*dk) f32 -> (B, T, H*dk).
    bf16 MFMA operands (performance mode only), no dropout, no gradient."""
    _f32(q, k, v)
    B, T, C = q.shape
    dk = C // H
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and klen.dtype == torch.int64
    o = torch.empty_like(q)
    call("osp_attn_fused_fwd", q, k, v, klen, o, B, H, T, dk, 1.0 / float(dk) ** 0.5)
    return o


def attn_train_fwd(q, k, v, klen, H, drop_p=0.0, seed=0, stream_id=0, sbias=None):
    """Fused training forward (csrc/attention_train.hip): (B, T, H*dk) f32 q, k, v -> (o, lse); the (T x T) scores never exist.
    The dropout mask is the one osp_attn_softmax_fwd would draw from the same (seed, stream_id)."""
    _f32(q, k, v)
    B, T, C = q.shape
    dk = C // H
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and klen.dtype == torch.int64
    o = torch.empty_like(q)
    lse = torch.empty((B * H, T), device=q.device, dtype=torch.float32)
    sh, sd = _seed(seed)
    if sbias is not None:
        _f32(sbias)
        assert sbias.is_contiguous() and sbias.numel() == B * H * T * T
    call("osp_attn_train_fwd", q, k, v, klen, sbias, o, lse, B, H, T, dk, 1.0 / float(dk) ** 0.5, float(drop_p), sh, sd, int(stream_id))
    return o, lse


def attn_train_bwd(q, k, v, o, lse, dout, klen, H, drop_p=0.0, seed=0, stream_id=0, sbias=None, want_dsbias=False):
    """Fused backward: recomputes the probabilities tile by tile; returns (dq, dk, dv) in the (B, T, H*dk) layout (+ the score
    term's gradient when ``sbias`` is given: (B*H, T, T), or None unless ``want_dsbias``)."""
    _f32(q, k, v, o, dout)
    B, T, C = q.shape
    dk = C // H
    assert dout.is_contiguous()
    dq, dkk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    dbuf = torch.empty_like(lse)
    sh, sd = _seed(seed)
    dsb = torch.empty_like(sbias) if (want_dsbias and sbias is not None) else None
    call("osp_attn_train_bwd", q, k, v, o, lse, dout, klen, sbias, dsb, dq, dkk, dv, dbuf, B, H, T, dk, 1.0 / float(dk) ** 0.5,
         float(drop_p), sh, sd, int(stream_id))
    return (dq, dkk, dv, dsb) if sbias is not None else (dq, dkk, dv)


def attn_softmax_bwd(P, dPd, scale, drop_p=0.0, seed=0, stream_id=0):
    """in place: dPd (gradient w.r.t. the dropped probabilities) -> gradient w.r.t. the raw scores."""
    T2 = P.shape[-1]
    sh, sd = _seed(seed)
    call("osp_attn_softmax_bwd", P, dPd, P.numel() // T2, T2, float(scale), float(drop_p), sh, sd, int(stream_id))
    return dPd
