"""autograd glue: each Function is a forward/backward *pair of HIP kernel sequences* behind the C ABI.

Parameter gradients are accumulated by the kernels directly into ``param.grad`` (the flat gradient
arena owned by the trainer, zeroed once per step); the Functions therefore return ``None`` for
parameter inputs.  Only activations flow through autograd.
"""
import torch

from . import kernels as K
from . import precision as _precision
from . import rng as _rng
from . import _lib


def gsink(p):
    """Gradient accumulation target of a parameter (allocated zero on first use)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def _want(p):
    return p is not None and p.requires_grad


# ------------------------------------------------------------------------------------------------ weight gradients off the chain
# The generator's backward is one dependent chain of SMALL kernels (50-600 workgroups on 256 CUs): input gradient of block n ->
# block n-1 -> ...  The weight / bias gradients hang off that chain -- nothing downstream reads them before the optimizer -- so
# they are issued on a side stream and fill the CUs the chain leaves idle.  Ordering: the side stream waits for the calling
# stream at every hand-over (the operands were just produced there); the engine's end-of-backward callback makes the stream
# that called backward() wait for the side streams, so whoever reads .grad afterwards (optimizer, all-reduce, a test) is
# ordered after them exactly as with inline launches.  Operands stay referenced until that join (the caching allocator would
# otherwise hand their memory to the next kernel of the calling stream while the side stream still reads it).
import os as _os

#: HIP priority of the generator-side helper streams (vocoder, weight-gradient side streams, CTC side stream)
_CHAIN_PRIORITY = int(_os.environ.get("OSP_PRIO_CHAIN", "0"))
_WG = {"on": _os.environ.get("OSP_WGRAD_STREAM", "1") != "0", "sides": {}, "used": [], "queued": False, "pending": [], "done": [],
       "regions": 0, "every": int(_os.environ.get("OSP_WGRAD_FLUSH", "1"))}


class _Side:
    """The weight-gradient stream of one calling stream, with its own hand-over events.  An event is only ever re-recorded on
    the SAME calling stream and only in a later backward pass (``pos`` restarts at the join), so a wait queued on the side stream
    can never come to refer to a record made on another stream (an earlier version shared one ring of 32 events between all
    calling streams and relied on hipStreamWaitEvent capturing the record at call time)."""
    __slots__ = ("stream", "raw", "events", "pos")

    def __init__(self, dev, name="wg_other"):
        from . import lanes
        self.stream = lanes.stream(name, torch.device("cuda", dev)) if lanes.managed() else torch.cuda.Stream(device=dev, priority=_CHAIN_PRIORITY)
        self.raw = self.stream.cuda_stream
        self.events, self.pos = [], 0

    def next_event(self):
        if self.pos == len(self.events):
            ev = torch.cuda.Event()
            ev.record()                                              # (torch creates the hipEvent lazily: force the handle to exist)
            self.events.append(ev)
        ev = self.events[self.pos]
        self.pos += 1
        return ev


def _flush_wgrad():
    """Launch the deferred weight-gradient calls on the side stream of the current stream, after everything queued on it so far."""
    pend = _WG["pending"]
    _WG["regions"] = 0
    if not pend:
        return
    dev = torch.cuda.current_device()
    raw = _lib._raw_stream(dev)
    side = _WG["sides"].get((dev, raw))
    if side is None:
        # (which calling stream this is decides the side stream's lane: the default stream = the acoustic model's, the vocoder's)
        vs = _named.get(("vocoder", dev))
        name = "wg_main" if raw == 0 else ("wg_voc" if (vs is not None and vs.cuda_stream == raw) else "wg_other")
        side = _WG["sides"][(dev, raw)] = _Side(dev, name)
    if side not in _WG["used"]:
        _WG["used"].append(side)
    ev = side.next_event()
    lib = _lib.lib()
    # record on the current stream (everything the operands depend on) + side stream waits: one C-ABI call on the raw handles
    lib.call("osp_stream_handover", ev.cuda_event, side.raw)
    _lib._STREAM_OVERRIDE[0] = side.raw
    try:
        for name, args in pend:
            lib.call(name, *args)
    finally:
        _lib._STREAM_OVERRIDE[0] = None
    # the argument tuples held the operands alive until here; from now on the side stream's queue order does: the join below
    # makes the calling stream wait before anything can reuse their memory (the list is cleared only after that wait is queued)
    _WG["done"].append(pend)
    _WG["pending"] = []


def _wait_for(side_raw, event):
    """The CURRENT stream waits for everything queued on the stream ``side_raw`` so far: ``event`` is recorded on the side stream and
    waited for here, in one C-ABI call (osp_stream_handover issued ON the side stream with this stream as its target) -- so that
    the join lands on a call tape like any other launch (optispeech_amd/tape.py)."""
    cur_raw = _lib._STREAM_OVERRIDE[0] or _lib._raw_stream(torch.cuda.current_device())
    keep = _lib._STREAM_OVERRIDE[0]
    _lib._STREAM_OVERRIDE[0] = side_raw
    try:
        _lib.lib().call("osp_stream_handover", event.cuda_event, cur_raw)
    finally:
        _lib._STREAM_OVERRIDE[0] = keep


def _join_wgrad():
    _flush_wgrad()
    for side in _WG["used"]:
        _wait_for(side.raw, side.next_event())
        side.pos = 0
    _WG["used"], _WG["queued"], _WG["done"] = [], False, []


def begin_backward():
    """Called by the step right before it runs backward(): if a previous backward pass raised (OOM, kernel error), the
    engine's end-of-backward callback never ran and the hand-over state is stale -- 'queued' still set (so no callback would be
    queued for THIS pass and its side-stream work would never be joined) and 'pending' holding launches of the failed pass.
    Drop the stale launches, join whatever did reach the side streams, start clean."""
    if _WG["queued"] or _WG["pending"] or _WG["used"]:
        _lib._RECORD[0] = None
        _WG["pending"] = []
        _WG["regions"] = 0
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            cur = torch.cuda.current_stream()
            for side in _WG["used"]:
                cur.wait_stream(side.stream)
                side.pos = 0
        _WG["used"], _WG["queued"], _WG["done"] = [], False, []


def nested_backward_begin():
    """A backward pass run INSIDE another one (tape.Segment records its inner graph's backward): the inner pass gets a clean
    hand-over state, so that it queues -- and records -- its own end-of-pass join instead of relying on the outer pass's."""
    outer = (_WG["queued"], _WG["pending"], _WG["used"], _WG["done"], _WG["regions"])
    _WG["queued"], _WG["pending"], _WG["used"], _WG["done"], _WG["regions"] = False, [], [], [], 0
    return outer


def nested_backward_end(outer):
    _WG["queued"], _WG["pending"], _WG["used"], _WG["done"], _WG["regions"] = outer


def wgrad_side_streams():
    """Side streams with weight-gradient work of the running backward pass in flight (dp.GradReducer orders a gradient-ready
    collective behind them: the ready signal comes from the calling stream, which does not wait for them before the join)."""
    return [s.stream for s in _WG["used"]]


class side_wgrad:
    """``with side_wgrad(t0, t1, ...):`` -- the enclosed C-ABI launches (weight-gradient kernels) are RECORDED: a kernels.* call
    inside may only allocate tensors that it passes to a recorded launch (the bf16 operand copies of kernels._bf16_rows: the argument
    tuple keeps them alive until the end-of-backward join; the split workspace is NOT such a tensor, it is one persistent buffer per
    launch stream, kernels.wgrad_workspace), never one it drops before the launch runs, and every OSP_WGRAD_FLUSH-th region the recorded launches go to the side stream of the
    current stream in one hand-over (one event record + one stream wait, torch's current stream is never switched).  Measured on
    one box, 40 steps each, twice: inline 21.84 / 20.75 ms per step, flush every region 20.22 / 20.42, every 2nd 20.92 / 20.88,
    every 4th 20.56 / 20.59, every 8th 20.41 / 20.57 -- the earlier start of the side-stream work is worth more than the host time
    of the hand-overs (~3 ms per step on an idle GPU), so the default is 1.  The engine's end-of-backward callback flushes the rest
    and makes the calling stream wait for the side stream.  Outside a backward pass and under hipGraph capture the launches stay
    inline."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.on = False

    def __enter__(self):
        if not _WG["on"] or not self.tensors or not self.tensors[0].is_cuda:
            return self
        if torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture the launches stay inline: the runtime executes a captured fork almost serially anyway
            # (DESIGN.md section 10), and ending the capture of the segmented step with such a fork in it crashed the runtime
            return self
        if not _WG["queued"]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_join_wgrad)
            except RuntimeError:                                   # not inside backward(): stay inline
                return self
            _WG["queued"] = True
        _lib._RECORD[0] = _WG["pending"]
        self.on = True
        return self

    def __exit__(self, *exc):
        if self.on:
            _lib._RECORD[0] = None
            _WG["regions"] += 1
            if _WG["regions"] >= _WG["every"]:
                _flush_wgrad()
        return False


_FUSED_LN_DW = _os.environ.get("OSP_FUSED_LN_DW", "1") != "0"
_FUSED_ATTN = _os.environ.get("OSP_FUSED_ATTN", "1") != "0"
_FUSED_ATTN_TRAIN = _os.environ.get("OSP_FUSED_ATTN_TRAIN", "1") != "0"


#: grad mode of the CALLER of the autograd Function being applied.  Inside Function.forward autograd is always off, and
#: ctx.needs_input_grad only repeats the inputs' requires_grad flags: a parameter fed to a Function under torch.no_grad() still says
#: "needs grad", so the synthesise path used to write every saved activation (LayerNorm statistics, pre-activations, z) it could
#: never use.  _grad_aware() wraps a Function's apply to record the outer mode; _saving(ctx) is what forwards ask.
_OUTER_GRAD = [True]


def _saving(ctx):
    return _OUTER_GRAD[0] and any(ctx.needs_input_grad) and not torch.is_inference_mode_enabled()


def _grad_aware(cls):
    inner = cls.apply

    def apply(*args):
        prev = _OUTER_GRAD[0]
        _OUTER_GRAD[0] = torch.is_grad_enabled()
        try:
            return inner(*args)
        finally:
            _OUTER_GRAD[0] = prev
    cls.apply = staticmethod(apply)
    return cls


#: the one-kernel MLP of a ConvNeXt block on the no-grad path (OSP_FUSED_MLP=0: the two conv-GEMM launches, for A/B measurements)
_FUSED_MLP = _os.environ.get("OSP_FUSED_MLP", "1") != "0"


@_grad_aware
class ConvNeXtBlockFn(torch.autograd.Function):
    """ConvNeXtBlock.forward + the backbone's per-block mask (generator/modules/convnext.py:34-47, :99-101).

    y = (x + rowscale * gamma * (W2 gelu(W1 LN(dwconv7(x)) + b1) + b2)) * rowmask
    x (B,T,C); dw (7,C) native tap-major; W1 (I,C); W2 (C,I); rowmask/rowscale (B*T,) or None.
    """

    @staticmethod
    def forward(ctx, x, dw, dwb, lnw, lnb, W1, b1, W2, b2, gamma, rowmask, rowscale, rowf_pre=None):
        B, T, C = x.shape
        I = W1.shape[0]
        M = B * T
        save = _saving(ctx)
        x = x.contiguous()
        ctx.lowp = _precision.is_bf16() and I % 64 == 0 and C % 64 == 0
        # lowp: h goes straight out as bf16 (its only consumers are the bf16 pointwise GEMM and its weight-gradient GEMM)
        h, xhat, rstd = K.dwconv7_ln_fwd(x, dw, dwb, lnw, lnb, 1e-6, save, h_bf16=ctx.lowp)
        h2 = h.view(M, C)
        z = torch.empty((M, C), device=x.device, dtype=torch.float32) if save else None
        if ctx.lowp and not save and _FUSED_MLP and K.mlp_fused_supported(C, I):
            # no gradient wanted (synthesise; the training step's decoder, which receives none): pwconv1 -> GELU -> pwconv2 -> gamma / residual / mask in ONE kernel, the (M, I)
            # hidden activations stay in registers (csrc/mlp_fused.hip)
            return K.convnext_mlp_fused(h2, W1, b1, W2, b2, gamma, x.view(M, C), rowmask, rowscale).view(B, T, C)
        if ctx.lowp:
            # performance mode: the I-wide intermediates (pre-activation u, gelu(u)) live in bf16 -- what autocast does
            # to these matmul outputs in the reference's default `16-mixed` precision; the block's input / output /
            # residual stream and the LayerNorm statistics stay f32.  Halves the dominant HBM traffic of the block and
            # puts pwconv2 on the direct-to-LDS kernel (bf16 A operand).
            u = torch.empty((M, I), device=x.device, dtype=torch.bfloat16) if save else None
            g = K.conv_gemm_bf16(h2, K.param_bf16(W1), I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=b1, aux_out=u,
                                 out_bf16=True)
            y = K.conv_gemm_bf16(g, K.param_bf16(W2), C, M=M, Trows=M, Tin=M, cin=I, epi=K.EPI_SCALE_RES_MASK, bias=b2,
                                 gamma=gamma, res=x.view(M, C), rowmask=rowmask, rowscale=rowscale, aux_out=z)
        else:
            u = torch.empty((M, I), device=x.device, dtype=torch.float32) if save else None
            g = K.conv_gemm(h2, W1, I, epi=K.EPI_GELU, bias=b1, aux_out=u)
            y = K.conv_gemm(g, W2, C, epi=K.EPI_SCALE_RES_MASK, bias=b2, gamma=gamma, res=x.view(M, C),
                            rowmask=rowmask, rowscale=rowscale, aux_out=z)
        if save:
            if rowf_pre is not None:
                rowf = rowf_pre                                  # rowmask * rowscale, computed for all blocks of the backbone at once
            elif rowmask is not None and rowscale is not None:
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
            with side_wgrad(dy2, z, rowf):
                K.colsum_prod(dy2, z, rowf, gsink(gamma))        # dgamma[c] += sum_m rowf[m] dy[m,c] z[m,c]   (one launch)
        # du[m,k] = rowf[m] * sum_n dy[m,n] * gamma[n] W2[n,k] * gelu'(u[m,k])
        # performance mode is decided HERE: a block whose forward ran in exact f32 (index-critical path, f32 u / g / h saved)
        # still takes the bf16 MFMA kernels for its gradients -- those feed no index
        if ctx.lowp or (_precision.is_bf16() and dy.is_cuda and I % 64 == 0 and C % 64 == 0):
            # gamma is folded into the transposed bf16 weight pack (kscale), the row factor into the bf16 copy of dy, which is
            # the A operand of both the input-gradient GEMM (direct-to-LDS kernels) and the weight-gradient GEMM
            dys = K.cast_bf16_rows(dy2, rowf)
            du = K.conv_gemm_bf16(dys, K.param_bf16_scaled_t(W2, gamma), I, M=M, Trows=M, Tin=M,
                                  cin=C, epi=K.EPI_GELU_BWD, aux_in=u, out_bf16=True)
            dh = K.conv_gemm_bf16(du, K.param_bf16(W1, transposed=True), C, M=M, Trows=M, Tin=M, cin=I)
            with side_wgrad(dys, g, du, h):
                if _want(W2):
                    K.conv_wgrad_bf16(dys, g, gsink(W2), gsink(b2) if _want(b2) else None, M=M, Trows=M, Tin=M, n=C, cin=I,
                                      oscale=gamma)
                if _want(W1):
                    K.conv_wgrad_bf16(du, h.view(M, C), gsink(W1), gsink(b1) if _want(b1) else None, M=M, Trows=M,
                                      Tin=M, n=I, cin=C)
        else:
            # exact-f32 path (f32 / "mixed" modes).  The transposed weights are materialised k-contiguous once per optimizer epoch:
            # as k-strided views the two input-gradient GEMMs stay on the register-staged kernel, as packs they take the
            # direct-to-LDS one (csrc/gemm_f32_glds.hip)
            packs = dy.is_cuda and K.f32_packs_ok()
            if packs:
                du = K.conv_gemm(dy2, K.param_f32_t(W2, gamma), I, cin=C, epi=K.EPI_GELU_BWD, rowscale=rowf, aux_in=u)
            else:
                du = K.conv_gemm(dy2, W2 * gamma[:, None], I, cin=C, w_strides=(1, 0, I), epi=K.EPI_GELU_BWD, rowscale=rowf, aux_in=u)
            if _want(W2):
                K.conv_wgrad(dy2, g, gsink(W2), gsink(b2) if _want(b2) else None, arow=rowf, oscale=gamma)
            dh = K.conv_gemm(du, K.param_f32_t(W1), C, cin=I) if packs else K.conv_gemm(du, W1, C, cin=I, w_strides=(1, 0, C))
            if _want(W1):
                K.conv_wgrad(du, h.view(M, C), gsink(W1), gsink(b1) if _want(b1) else None)
        wl, wd = _want(lnw), _want(dw)
        if C <= 384 and dy.is_cuda and _FUSED_LN_DW:
            # one pass: the LayerNorm input gradient dc never goes to HBM (csrc/convnext.hip: ln_dwconv7_bwd_kernel)
            dx = K.ln_dwconv7_bwd(dh, xhat.view(M, C), rstd.view(M), lnw, x, dw, dy2.view(B, T, C), rowmask,
                                  gsink(lnw) if wl else None, gsink(lnb) if wl else None, gsink(dw) if wd else None,
                                  gsink(dwb) if wd else None)
            return (dx,) + (None,) * 12
        dc = K.layernorm_bwd(dh, xhat.view(M, C), None, rstd.view(M), lnw, gsink(lnw) if wl else None,
                             gsink(lnb) if wl else None)
        dx = K.dwconv7_bwd(dc.view(B, T, C), x, dw, dy2.view(B, T, C), rowmask, gsink(dw) if wd else None,
                           gsink(dwb) if wd else None)
        return (dx,) + (None,) * 12


@_grad_aware
class LayerNormFn(torch.autograd.Function):
    """y = (LN_C(x) * w + b) * dropout * rowmask  (nn.LayerNorm call sites convnext.py:102, wavenext:84)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, rowmask, drop_p, seed, stream_id):
        x = x.contiguous()
        save = _saving(ctx)
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


# ------------------------------------------------------------------------------------------------ dense conv / linear
@_grad_aware
class ConvLinearFn(torch.autograd.Function):
    """y = act(conv1d_k(x) + b) on channels-last frames (nn.Conv1d / nn.Linear call sites of the path).

    x (B,T,Cin); w: the *parameter itself* holding Cout*taps*Cin floats in native (Cout, taps, Cin) order (any
    view shape -- it must be the leaf so that its gradient arena slot is found); act in {None, "relu"};
    rowmask (B*T,) optional output mask.
    """

    @staticmethod
    def forward(ctx, x, w, b, cout, taps, pad, act, rowmask, out_bf16=False):
        B, T, Cin = x.shape
        assert w.numel() == cout * taps * Cin and w.is_contiguous(), (tuple(w.shape), cout, taps, Cin)
        x = x.contiguous()
        epi = K.EPI_RELU if act == "relu" else (K.EPI_MASK if rowmask is not None else K.EPI_NONE)
        assert not (act == "relu" and rowmask is not None)
        if (not _saving(ctx)) and x.is_cuda and _precision.is_bf16() and Cin % 64 == 0 and cout % 64 == 0 and B * T >= 8192:
            # no gradient wanted and a large row count (synthesise: the vocoder's embedding conv and head at ~50 k frames): a bf16
            # copy of the rows puts the GEMM on the direct-to-LDS kernels (an f32 A operand takes the register-staged kernel, which
            # converts behind every load: 196 vs ~110 us per launch here); `out_bf16` hands the next conv_linear its operand as is.
            # Same values either way: the loaders round an f32 A operand to bf16 in the same place.
            xb = x if x.dtype == torch.bfloat16 else K.cast_bf16(x)
            y = K.conv_gemm_bf16(xb.view(B * T, Cin), K._param_pack(w, w, cout, taps, Cin, (taps * Cin, Cin, 1)), cout, M=B * T,
                                 Trows=T, Tin=T, cin=Cin, taps=taps, a_off=-pad, epi=epi, bias=b, rowmask=rowmask, out_bf16=out_bf16)
            return y.view(B, T, cout)
        if x.dtype != torch.float32:                                  # a bf16 hand-over whose consumer is not on the path above (odd widths)
            x = x.float()
        y = K.conv_gemm(x.view(B * T, Cin), w, cout, T=T, taps=taps, pad=pad, bias=b, epi=epi, rowmask=rowmask, w_param=w)
        if _saving(ctx):
            ctx.save_for_backward(x, y if act == "relu" else None, rowmask)
            ctx.params = (w, b)
            ctx.cfg = (cout, taps, pad, act)
        return y.view(B, T, cout)

    @staticmethod
    def backward(ctx, dy):
        x, y, rowmask = ctx.saved_tensors
        w, b = ctx.params
        Cout, taps, pad, act = ctx.cfg
        B, T, Cin = x.shape
        g = dy.contiguous().view(B * T, Cout)
        if act == "relu":
            g = K.relu_mask(g, y) if (g.is_cuda and g.dtype == torch.float32) else g * (y > 0)
        # dgrad: dx[m,c] = sum_{j,n} g[m - j + pad, n] w[n, j, c]  == conv with flipped taps over the (n) axis
        dx = None
        if ctx.needs_input_grad[0]:
            base = w.detach().view(Cout, taps, Cin)[:, taps - 1:, :]          # pointer to the last tap
            dx = K.conv_gemm(g, base, Cin, T=T, taps=taps, pad=taps - 1 - pad, cin=Cout,
                             w_strides=(1, -Cin, taps * Cin), a_rowscale=rowmask, w_param=w).view(B, T, Cin)
        if _want(w):
            with side_wgrad(g, x, rowmask):
                K.conv_wgrad(g, x.view(B * T, Cin), gsink(w), gsink(b) if _want(b) else None, T=T, taps=taps, pad=pad,
                             arow=rowmask)
        return (dx,) + (None,) * 8


@_grad_aware
class LSTMFn(torch.autograd.Function):
    """hs = nn.LSTM(H, H, num_layers=1, batch_first=True)(x)[0] with zero initial state (leanspeech.py:49-60).

    x (B, T, H); w_ih, w_hh (4H, H), b_ih, b_hh (4H,): torch's parameters (gate order i, f, g, o).  Input projection, input
    gradient and the three weight gradients are GEMMs of the conv-GEMM family; the recurrence runs in csrc/lstm.hip."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        B, T, H = x.shape
        x = x.contiguous()
        save = _saving(ctx)
        gx = K.conv_gemm(x.view(B * T, H), w_ih, 4 * H, bias=b_ih + b_hh).view(B, T, 4 * H)
        hs, gates, cs = K.lstm_fwd(gx, w_hh.detach().contiguous(), save=save)
        if save:
            ctx.save_for_backward(x, hs, gates, cs)
            ctx.params = (w_ih, w_hh, b_ih, b_hh)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        x, hs, gates, cs = ctx.saved_tensors
        w_ih, w_hh, b_ih, b_hh = ctx.params
        B, T, H = x.shape
        dg = K.lstm_bwd(dhs.contiguous(), gates, cs, w_hh.detach().contiguous())
        dg2 = dg.view(B * T, 4 * H)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.conv_gemm(dg2, w_ih.detach(), H, cin=4 * H, w_strides=(1, 0, H)).view(B, T, H)      # dg @ W_ih
        h_prev = torch.cat([hs.new_zeros((B, 1, H)), hs[:, :-1]], 1).contiguous()                      # h_{t-1}, zero initial state
        with side_wgrad(dg2, x, h_prev):
            if _want(w_ih):
                K.conv_wgrad(dg2, x.view(B * T, H), gsink(w_ih), gsink(b_ih) if _want(b_ih) else None)
            if _want(w_hh):
                K.conv_wgrad(dg2, h_prev.view(B * T, H), gsink(w_hh), gsink(b_hh) if _want(b_hh) else None)
        return dx, None, None, None, None


def lstm(x, w_ih, w_hh, b_ih, b_hh):
    return LSTMFn.apply(x, w_ih, w_hh, b_ih, b_hh)


@_grad_aware
class DepthwiseConvFn(torch.autograd.Function):
    """y = depthwise_conv_K(x) (+ bias) on channels-last frames (nn.Conv1d(C, C, K, padding=K//2, groups=C) call sites:
    ConvSeparable modules/layers.py:455-477).  x (B,T,C); w (K,C) tap-major parameter; bias (C,) or None."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        y = K.dwconv_fwd(x, w, bias)
        if _saving(ctx):
            ctx.save_for_backward(x)
            ctx.params = (w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, bias = ctx.params
        dy = dy.contiguous()
        dx = K.dwconv_fwd(dy, w, None, None, flip=True) if ctx.needs_input_grad[0] else None
        if _want(w):
            with side_wgrad(dy, x):
                K.dwconv_wgrad(dy, x, gsink(w), gsink(bias) if _want(bias) else None)
        return dx, None, None


def depthwise_conv(x, w, bias=None):
    return DepthwiseConvFn.apply(x, w, bias)


class DropoutAddFn(torch.autograd.Function):
    """res + dropout(x)  (F.dropout + residual sites of the separable-conv / Conformer layers); res may be None."""

    @staticmethod
    def forward(ctx, x, res, p, seed, stream_id):
        ctx.cfg = (p, seed, stream_id)
        ctx.has_res = res is not None
        return K.dropout_add(x.contiguous(), p, seed, stream_id, None if res is None else res.contiguous())

    @staticmethod
    def backward(ctx, dy):
        p, seed, stream_id = ctx.cfg
        dy = dy.contiguous()
        dx = K.dropout_add(dy, p, seed, stream_id) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.has_res and ctx.needs_input_grad[1] else None), None, None, None


def dropout_add(x, p, training, stream_id, res=None):
    """res + F.dropout(x, p, training) with the package's counter-based RNG."""
    if not training or p <= 0.0:
        return x if res is None else x + res
    return DropoutAddFn.apply(x, res, float(p), _rng.seed(), stream_id)


def conv_linear(x, w, b, cout, taps=1, pad=0, act=None, rowmask=None, out_bf16=False):
    """out_bf16: the caller feeds the result to another conv_linear and wants no gradient (it is honoured only on the no-grad
    large-row path; otherwise the result is f32 as always)."""
    return ConvLinearFn.apply(x, w, b, cout, taps, pad, act, rowmask, out_bf16 and not torch.is_grad_enabled())


@_grad_aware
class PredictorLayerFn(torch.autograd.Function):
    """One VariancePredictor layer: Conv1d(k) -> ReLU -> LayerNorm(C, eps 1e-12) -> Dropout (core.py:62-76)."""

    @staticmethod
    def forward(ctx, x, w, b, lnw, lnb, taps, drop_p, seed, stream_id):
        B, T, Cin = x.shape
        Cout = w.shape[0]
        pad = (taps - 1) // 2
        x = x.contiguous()
        r = K.conv_gemm(x.view(B * T, Cin), w, Cout, T=T, taps=taps, pad=pad, bias=b, epi=K.EPI_RELU, w_param=w)
        save = _saving(ctx)
        y, mean, rstd = K.layernorm_fwd(r, lnw, lnb, 1e-12, save=save, drop_p=drop_p, seed=seed, stream_id=stream_id)
        if save:
            ctx.save_for_backward(x, r, mean, rstd)
            ctx.params = (w, b, lnw, lnb)
            ctx.cfg = (taps, pad, drop_p, seed, stream_id)
        return y.view(B, T, Cout)

    @staticmethod
    def backward(ctx, dy):
        x, r, mean, rstd = ctx.saved_tensors
        w, b, lnw, lnb = ctx.params
        taps, pad, drop_p, seed, stream_id = ctx.cfg
        B, T, Cin = x.shape
        Cout = w.shape[0]
        wl = _want(lnw)
        g = K.layernorm_bwd(dy.contiguous().view(B * T, Cout), r, mean, rstd, lnw, gsink(lnw) if wl else None,
                            gsink(lnb) if wl else None, relu_src=r, drop_p=drop_p, seed=seed, stream_id=stream_id)
        dx = None
        if ctx.needs_input_grad[0]:
            base = w.detach()[:, taps - 1:, :]
            dx = K.conv_gemm(g, base, Cin, T=T, taps=taps, pad=taps - 1 - pad, cin=Cout,
                             w_strides=(1, -Cin, taps * Cin), w_param=w).view(B, T, Cin)
        if _want(w):
            with side_wgrad(g, x):
                K.conv_wgrad(g, x.view(B * T, Cin), gsink(w), gsink(b) if _want(b) else None, T=T, taps=taps, pad=pad)
        return (dx,) + (None,) * 8


@_grad_aware
class VarianceEmbedFn(torch.autograd.Function):
    """x_out = (x + dropout(Conv1d(1 -> C, k)(values))) * keep_mask   (core.py:161-165 / :170-175)."""

    @staticmethod
    def forward(ctx, x, values, w, b, rowmask, drop_p, seed, stream_id):
        B, T, C = x.shape
        taps = w.shape[1]
        pad = (taps - 1) // 2
        x = x.contiguous()
        v = values.contiguous().view(B * T, 1)
        if drop_p > 0.0:
            # x + dropout(emb) in one launch; the backward regenerates the mask from the same Philox counter (element index)
            emb = K.conv_gemm(v, w, C, T=T, taps=taps, pad=pad, bias=b)
            y = K.dropout_add(emb, drop_p, seed, stream_id, res=x.view(B * T, C))
            y = K.ew_mul_rows(y, rowmask, out=y) if y.is_cuda else y * rowmask[:, None]
            dm = None
        else:
            dm = None
            y = K.conv_gemm(v, w, C, T=T, taps=taps, pad=pad, bias=b, epi=K.EPI_SCALE_RES_MASK, res=x.view(B * T, C),
                            rowmask=rowmask)
        if _saving(ctx):
            ctx.save_for_backward(v, rowmask, dm)
            ctx.params = (w, b)
            ctx.cfg = (B, T, C, taps, pad)
            ctx.drop = (drop_p, seed, stream_id)
        return y.view(B, T, C)

    @staticmethod
    def backward(ctx, dy):
        v, rowmask, dm = ctx.saved_tensors
        w, b = ctx.params
        B, T, C, taps, pad = ctx.cfg
        g = dy.contiguous().view(B * T, C)
        g = K.ew_mul_rows(g, rowmask) if g.is_cuda else g * rowmask[:, None]
        dx = g.view(B, T, C) if ctx.needs_input_grad[0] else None
        drop_p, seed, stream_id = ctx.drop
        ge = K.dropout_add(g, drop_p, seed, stream_id) if drop_p > 0.0 else g
        dval = None
        if ctx.needs_input_grad[1]:
            base = w.detach()[:, taps - 1:, :]
            dval = K.conv_gemm(ge, base, 1, T=T, taps=taps, pad=taps - 1 - pad, cin=C,
                               w_strides=(1, -1, taps)).view(B, T)
        if _want(w):
            K.conv_wgrad(ge, v, gsink(w), gsink(b) if _want(b) else None, T=T, taps=taps, pad=pad)
        return dx, dval, None, None, None, None, None, None


def dropout_mask(shape, p, seed, stream_id, device):
    """Materialised keep/scale mask from the same Philox stream the fused kernels use (cold paths only)."""
    ones = torch.ones(shape, device=device, dtype=torch.float32)
    w = torch.ones(shape[-1], device=device)
    b = torch.zeros(shape[-1], device=device)
    # LN of a constant row is 0*w+b; use b=1 trick: y = (0*1 + 1) * dropout
    y, _, _ = K.layernorm_fwd(ones, w, b + 1.0, 1.0, save=False, drop_p=p, seed=seed, stream_id=stream_id)
    return y


# ------------------------------------------------------------------------------------------------ text embedding
@_grad_aware
class TextEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, E, scale, pos, drop_p, seed, stream_id):
        out = K.text_embed_fwd(tok, E, pos, scale, drop_p, seed, stream_id)
        if _saving(ctx):
            ctx.save_for_backward(tok, pos)
            ctx.params = (E, scale)
            ctx.cfg = (drop_p, seed, stream_id)
        return out

    @staticmethod
    def backward(ctx, dy):
        tok, pos = ctx.saved_tensors
        E, scale = ctx.params
        drop_p, seed, stream_id = ctx.cfg
        K.text_embed_bwd(dy, tok, pos, gsink(E) if _want(E) else None, gsink(scale) if _want(scale) else None,
                         0, drop_p, seed, stream_id)
        return (None,) * 7


# ------------------------------------------------------------------------------------------------ alignment
@_grad_aware
class AlignLogProbFn(torch.autograd.Function):
    """log_p_attn = log_softmax_n(-||f_t - e_n||_2 masked) + prior   (alignments.py:66-81)."""

    @staticmethod
    def forward(ctx, f, e, prior, x_len, y_len):
        f, e = f.contiguous(), e.contiguous()
        score = K.pairwise_score(f, e, x_len)
        lp, lse = K.logsoftmax_prior_fwd(score, prior)
        if _saving(ctx):
            ctx.save_for_backward(f, e, score, lse, x_len, y_len)
        return lp

    @staticmethod
    def backward(ctx, dlp):
        f, e, score, lse, x_len, y_len = ctx.saved_tensors
        B, T, C = f.shape
        N = e.shape[1]
        w, wrow = K.logsoftmax_prior_bwd(dlp, score, lse, x_len, y_len)
        # df = rowsum(w) * f - w @ e       (batched, NN mode, fused in the epilogue)
        df = torch.empty_like(f)
        K.conv_gemm(w, e, C, cin=N, w_strides=(1, 0, C), out=df, epi=K.EPI_AXMY, rowscale=wrow.view(-1), aux_in=f,
                    batch=B, batch_strides=(T * N, N * C, T * C, T * C))
        # de = colsum(w) * e - w^T @ f     (batched wgrad form; its bias-gradient output is colsum(w))
        wtf = torch.zeros((B, N, C), device=f.device, dtype=torch.float32)
        wcol = torch.zeros((B, N), device=f.device, dtype=torch.float32)
        K.conv_wgrad(w, f, wtf, wcol, batch=B)
        # de = colsum(w) * e - w^T f
        de = K.ew_mul_rows(e.view(B * N, C), wcol.view(-1)).view(B, N, C)
        K.ew_axpby(de, wtf, 1.0, -1.0, out=de)
        return df, de, None, None, None


_side_streams = {}
_pending_side = []


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _side_streams:
        from . import lanes
        _side_streams[key] = lanes.stream("ctc", torch.device("cuda", key)) if lanes.managed() else torch.cuda.Stream(device=key, priority=_CHAIN_PRIORITY)
    return _side_streams[key]


_side_evs = {}


def _side_events(device):
    """Two persistent events of the CTC side stream (fork, join), created with their handles (torch creates them lazily)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    ev = _side_evs.get(key)
    if ev is None:
        ev = (torch.cuda.Event(), torch.cuda.Event())
        for e in ev:
            e.record()
        _side_evs[key] = ev
    return ev


def join_side_stream():
    """Make the current stream wait for everything queued on the side stream so far (no host sync)."""
    while _pending_side:
        side, ev = _pending_side.pop()
        _wait_for(side.cuda_stream, ev)


class AlignLossFn(torch.autograd.Function):
    """(forwardsum_loss, bin_loss) from log_p_attn: ForwardSumLoss (loss.py:150-194) and the binarisation term of
    viterbi_decode (alignments.py:236-238); `path`/`bin_item` come from the MAS kernel."""

    @staticmethod
    def forward(ctx, lp, x_len, y_len, path, bin_item):
        B = lp.shape[0]
        need = ctx.needs_input_grad[0]
        lp = lp.contiguous()
        # The forward-sum recursion is latency-bound (2*T_mel dependent steps on one workgroup per utterance, ~1.3 ms at
        # B=32) and occupies 32 of the 256 CUs: it runs on a side stream next to the decoder / vocoder forward.  The
        # caller joins with ``join_side_stream()`` before it consumes the loss (generator.forward does, right before the
        # loss sum); the gradient saved here is only read in backward, i.e. after that join.
        side = _side_stream(lp.device)
        ev = _side_events(lp.device)
        # fork through the C ABI on raw handles (a call tape records it); the launches run with the side stream CURRENT, so that
        # everything they allocate -- including the recursion's workspace, which dies inside K.forwardsum_ctc -- belongs to the
        # side stream's pool: allocated under the calling stream it would be handed to the calling stream's next allocation while
        # the recursion is still writing it (found as non-finite duration-predictor gradients in round 4)
        _lib.lib().call("osp_stream_handover", ev[0].cuda_event, side.cuda_stream)
        main = torch.cuda.current_stream()
        with torch.cuda.stream(side):
            loss_item, grad = K.forwardsum_ctc(lp, x_len, y_len, want_grad=need)
            fs = K.sum_scaled(loss_item, 1.0 / B)
        for t in (lp, x_len, y_len):
            t.record_stream(side)
        for t in (fs, grad):
            if t is not None:
                t.record_stream(main)
        _pending_side.append((side, ev[1]))
        if need:
            ctx.save_for_backward(grad, path, y_len)
        return fs, K.sum_scaled(bin_item, 1.0 / B)

    @staticmethod
    def backward(ctx, g_fs, g_bin):
        grad, path, y_len = ctx.saved_tensors
        dlp = K.ew_scale_dev(grad, g_fs.reshape(1))
        K.bin_loss_bwd(path, y_len, g_bin.reshape(1).contiguous().float(), dlp)
        return dlp, None, None, None, None


class VarianceLossFn(torch.autograd.Function):
    """(duration, pitch, energy) losses of FastSpeech2Loss (loss.py:83-140)."""

    @staticmethod
    def forward(ctx, d_hat, p_hat, e_hat, ds, ps, es, x_len):
        out, gd, gp, ge = K.variance_losses(d_hat, p_hat, e_hat, ds, ps, es, x_len)
        ctx.save_for_backward(gd, gp, ge)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        gd, gp, ge = ctx.saved_tensors
        return (K.ew_scale_dev(gd, g0.reshape(1)), K.ew_scale_dev(gp, g1.reshape(1)), K.ew_scale_dev(ge, g2.reshape(1)),
                None, None, None, None)


class GaussianUpsampleFn(torch.autograd.Function):
    """hs' = softmax_n(-delta (t - c_n)^2) @ hs  (alignments.py:136-174); durations carry no gradient."""

    @staticmethod
    def forward(ctx, hs, ds, x_len, y_len, Tm, delta):
        B, N, C = hs.shape
        _, _, centre = K.duration_stats(ds, want_centre=True)
        P = K.gaussian_weights(centre, x_len, y_len, Tm, delta)
        hs = hs.contiguous()
        y = K.conv_gemm(P, hs, C, cin=N, w_strides=(1, 0, C), batch=B, batch_strides=(Tm * N, N * C, Tm * C, 0))
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(P)
            ctx.shape = (B, N, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        (P,) = ctx.saved_tensors
        B, N, C = ctx.shape
        dhs = torch.zeros((B, N, C), device=dy.device, dtype=torch.float32)
        K.conv_wgrad(P, dy.contiguous(), dhs, None, batch=B)
        return dhs, None, None, None, None, None


@_grad_aware
class AttentionFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(d_k) over the valid keys) V per head: MultiHeadedAttention.forward without the four linear layers
    (generator/modules/_transformer/attention.py:50-125).  q, k, v: (B, T, H * d_k) f32; klen (B,) int64 valid key counts.
    The batched GEMMs (batch = B * H) run on the conv-GEMM / wgrad kernels, the masked softmax + dropout on
    osp_attn_softmax_fwd/bwd; the (B, H, T, T) probabilities are kept for the backward (164 MB per decoder layer at B = 32,
    T = 800 -- a flash-style fused kernel is the round-2 item for this variant)."""

    @staticmethod
    def forward(ctx, q, k, v, klen, H, drop_p, seed, stream_id, sbias=None):
        """sbias (B * H, T, T) or None: added to Q K^T before the scaling (the relative-position term of
        RelPositionMultiHeadedAttention, _transformer/attention.py:290-313); it receives the score gradient."""
        B, T, C = q.shape
        dk = C // H
        Z = B * H
        if (_FUSED_ATTN and sbias is None and drop_p == 0.0 and not _saving(ctx) and q.is_cuda
                and _precision.is_bf16() and dk in (32, 64, 128)):
            # no-grad / inference: one flash-style kernel, the (B*H, T, T) scores never exist (csrc/attention.hip)
            return K.attn_fused_fwd(q.contiguous(), k.contiguous(), v.contiguous(), klen, H)
        ctx.fused = False
        if (_FUSED_ATTN_TRAIN and q.is_cuda and _precision.is_bf16() and dk in (32, 64, 128)):
            # training, performance mode: fused forward that keeps the per-row log-sum-exp, fused recomputing backward
            # (csrc/attention_train.hip) -- no (B*H, T, T) tensor in either direction, no head-major copies
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            sb = None if sbias is None else sbias.reshape(Z, T, T).contiguous()      # relative-position score term (Conformer)
            o, lse = K.attn_train_fwd(q, k, v, klen, H, drop_p, seed, stream_id, sbias=sb)
            if any(ctx.needs_input_grad[:3]) or (sb is not None and ctx.needs_input_grad[8]):
                ctx.save_for_backward(q, k, v, o, lse, klen, *([] if sb is None else [sb]))
                ctx.cfg = (H, drop_p, seed, stream_id)
                ctx.fused = True
                ctx.bias_shape = None if sbias is None else sbias.shape
            return o
        if q.is_cuda and dk % 4 == 0:                            # head split / merge as launches of ours (a call tape can hold them)
            heads = lambda t: K.permute_0213(t.contiguous().view(B, T, H, dk)).view(Z, T, dk)     # noqa: E731
        else:
            heads = lambda t: t.view(B, T, H, dk).permute(0, 2, 1, 3).contiguous().view(Z, T, dk)      # noqa: E731
        qh, kh, vh = heads(q), heads(k), heads(v)
        scale = 1.0 / float(dk) ** 0.5
        S = K.conv_gemm(qh, kh, T, cin=dk, w_strides=(dk, 0, 1), batch=Z, batch_strides=(T * dk, T * dk, T * T, 0))
        if sbias is not None:
            S.add_(sbias.reshape(Z, T, T))
        ctx.has_bias = sbias is not None
        P, Pd = K.attn_softmax_fwd(S, klen, B, H, T, T, scale, drop_p, seed, stream_id)
        O = K.conv_gemm(Pd, vh, dk, cin=T, w_strides=(1, 0, dk), batch=Z, batch_strides=(T * T, T * dk, T * dk, 0))
        if any(ctx.needs_input_grad[:3]) or (sbias is not None and ctx.needs_input_grad[8]):
            ctx.save_for_backward(qh, kh, vh, P, Pd)
            ctx.cfg = (B, T, H, dk, scale, drop_p, seed, stream_id)
        if O.is_cuda and dk % 4 == 0:
            return K.permute_0213(O.view(B, H, T, dk)).view(B, T, C)
        return O.view(B, H, T, dk).permute(0, 2, 1, 3).reshape(B, T, C)

    @staticmethod
    def backward(ctx, dout):
        if ctx.fused:
            saved = ctx.saved_tensors
            q, k, v, o, lse, klen = saved[:6]
            sb = saved[6] if len(saved) > 6 else None
            H, drop_p, seed, stream_id = ctx.cfg
            res = K.attn_train_bwd(q, k, v, o, lse, dout.contiguous(), klen, H, drop_p, seed, stream_id, sbias=sb,
                                   want_dsbias=sb is not None and ctx.needs_input_grad[8])
            dsb = res[3].view(ctx.bias_shape) if (sb is not None and res[3] is not None) else None
            return res[0], res[1], res[2], None, None, None, None, None, dsb
        qh, kh, vh, P, Pd = ctx.saved_tensors
        B, T, H, dk, scale, drop_p, seed, stream_id = ctx.cfg
        Z = B * H
        gpu = dout.is_cuda and dk % 4 == 0
        dO = (K.permute_0213(dout.contiguous().view(B, T, H, dk)) if gpu else dout.view(B, T, H, dk).permute(0, 2, 1, 3).contiguous()).view(Z, T, dk)
        dPd = K.conv_gemm(dO, vh, T, cin=dk, w_strides=(dk, 0, 1), batch=Z, batch_strides=(T * dk, T * dk, T * T, 0))
        dV = torch.zeros((Z, T, dk), device=dout.device, dtype=torch.float32)
        K.conv_wgrad(Pd, dO, dV, None, batch=Z)                          # dV[t2, d] = sum_t1 Pd[t1, t2] dO[t1, d]
        dS = K.attn_softmax_bwd(P, dPd, scale, drop_p, seed, stream_id)
        dQ = K.conv_gemm(dS, kh, dk, cin=T, w_strides=(1, 0, dk), batch=Z, batch_strides=(T * T, T * dk, T * dk, 0))
        dKh = torch.zeros((Z, T, dk), device=dout.device, dtype=torch.float32)
        K.conv_wgrad(dS, qh, dKh, None, batch=Z)                         # dK[t2, d] = sum_t1 dS[t1, t2] q[t1, d]
        if gpu:
            back = lambda t: K.permute_0213(t.view(B, H, T, dk)).view(B, T, H * dk)                 # noqa: E731
        else:
            back = lambda t: t.view(B, H, T, dk).permute(0, 2, 1, 3).reshape(B, T, H * dk)          # noqa: E731
        return back(dQ), back(dKh), back(dV), None, None, None, None, None, (dS if ctx.has_bias else None)


@_grad_aware
class ScaledPosEncFn(torch.autograd.Function):
    """x + alpha * pe[:T] over the batch (ScaledPositionalEncoding, _transformer/embedding.py:120-124) with alpha read on the device.
    dx = dy; d alpha = sum dy * pe (two small launches, fixed summation order)."""

    @staticmethod
    def forward(ctx, x, alpha, pe):
        x = x.contiguous()
        if ctx.needs_input_grad[1]:
            ctx.save_for_backward(pe)
        return K.posenc_fwd(x, pe, alpha)

    @staticmethod
    def backward(ctx, dy):
        dalpha = None
        if ctx.needs_input_grad[1]:
            (pe,) = ctx.saved_tensors
            dalpha = K.posenc_dalpha(dy.contiguous(), pe)
        return (dy if ctx.needs_input_grad[0] else None), dalpha, None


class BatchedNTFn(torch.autograd.Function):
    """C[z] = A[z] B[z]^T for z < Z on the conv-GEMM kernels: A (Z, M, K), B (Z, N, K) -> (Z, M, N), with both gradients
    (torch.matmul(x, y.transpose(-2, -1)) call sites of the relative-position attention)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        Z, M, Kd = a.shape
        N = b.shape[1]
        c = K.conv_gemm(a, b, N, cin=Kd, w_strides=(Kd, 0, 1), batch=Z, batch_strides=(M * Kd, N * Kd, M * N, 0))
        ctx.save_for_backward(a, b)
        return c

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        Z, M, Kd = a.shape
        N = b.shape[1]
        dc = dc.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:                                       # dA = dC B
            da = K.conv_gemm(dc, b, Kd, cin=N, w_strides=(1, 0, Kd), batch=Z, batch_strides=(M * N, N * Kd, M * Kd, 0))
        if ctx.needs_input_grad[1]:                                       # dB[n, k] = sum_m dC[m, n] A[m, k]
            db = torch.zeros((Z, N, Kd), device=dc.device, dtype=torch.float32)
            K.conv_wgrad(dc, a, db, None, batch=Z)
        return da, db


class JoinStreamAtBackwardEndFn(torch.autograd.Function):
    """Identity.  Placed on the output of a sub-graph that ran on a side stream: autograd replays that sub-graph's backward on
    the same side stream, and because parameter gradients are written straight into the gradient arena (no AccumulateGrad
    node) the engine would not make the stream backward() was called from wait for it -- this node's backward (the first
    node of the sub-graph's backward) queues that wait as an engine callback."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        side = torch.cuda.current_stream()
        torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream().wait_stream(side))
        return g


def run_on_side_stream(key, fn, inputs):
    """fn() on a persistent side stream ordered after the current stream; the current stream waits for it right away (the
    forward stays serial) -- the point is that autograd replays fn's backward on that stream, concurrently with whatever
    backward work runs on the current stream."""
    main = torch.cuda.current_stream()
    st = _side_stream(inputs[0].device) if key is None else _named_stream(key, inputs[0].device)
    st.wait_stream(main)
    with torch.cuda.stream(st):
        out = JoinStreamAtBackwardEndFn.apply(fn())
    for t in inputs:
        t.record_stream(st)
    main.wait_stream(st)
    out.record_stream(main)
    return out


class ClipFn(torch.autograd.Function):
    """clip(x, lo, hi) (WaveNeXtHead, wavenext/__init__.py:47) on osp_clip; the gradient passes where lo <= x <= hi."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        x = x.contiguous()
        out = torch.empty_like(x)
        K.call("osp_clip", x, None, out, x.numel(), float(lo), float(hi))
        ctx.save_for_backward(x)
        ctx.lim = (float(lo), float(hi))
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        K.call("osp_clip", x, dy.contiguous(), dx, x.numel(), ctx.lim[0], ctx.lim[1])
        return dx, None, None


_named = {}


def _named_stream(key, device):
    k = (key, torch.device(device).index)
    if k not in _named:
        from . import lanes
        from .model.discriminator import _skew
        _skew(device, "OSP_SKEW_" + str(key).upper())
        _named[k] = (lanes.stream("voc", device) if (lanes.managed() and key == "vocoder")
                     else torch.cuda.Stream(device=device, priority=_CHAIN_PRIORITY))
    return _named[k]


# ------------------------------------------------------------------------------------------------ scalar loss assembly
class WeightedSumFn(torch.autograd.Function):
    """sum_i coeff_i * term_i over device scalars in ONE autograd node and one launch each way (osp_dot_multi / osp_scale_vec; the
    gradients of the terms are views of one vector).  The loss assemblies (generator/__init__.py:175-181,
    vocoder/wavenext/disc/__init__.py:105-111) were ~8 tiny launches forward and as many backward each."""

    @staticmethod
    def forward(ctx, coeffs, *terms):
        ctx.coeffs = coeffs
        ts = [t.reshape(()) if t.dtype == torch.float32 else t.reshape(()).float() for t in terms]
        if ts[0].is_cuda:
            return K.dot_multi(ts, coeffs)
        return torch.dot(torch.stack(ts), torch.tensor(coeffs, dtype=torch.float32))

    @staticmethod
    def backward(ctx, g):
        if g.is_cuda:
            gs = K.scale_vec(g.reshape(1).float(), ctx.coeffs)
        else:
            gs = g * torch.tensor(ctx.coeffs, dtype=torch.float32)
        return (None,) + tuple(gs.unbind(0))


def weighted_sum(terms, coeffs):
    """terms: device scalars, coeffs: python floats."""
    return WeightedSumFn.apply(tuple(float(c) for c in coeffs), *terms)
