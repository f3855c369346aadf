"""Multi-period discriminator stacks on the bf16-MFMA conv-GEMM.

DiscriminatorP (vocoder/wavenext/disc/_discriminators.py:41-97) applies Conv2d((k,1), stride (s,1)) over the
period-folded wave (B, 1, T/p, p): every (utterance, period column) is an independent 1-D sequence.  In channels-last
layout (U = B*p sequences, T frames, C channels) each layer is a strided k-tap Conv1d = one GEMM, LeakyReLU fused in
the epilogue, activations kept in bf16.  The whole stack is ONE autograd Function so the backward can fuse
(dgrad + incoming feature-matching gradient) * LeakyReLU' into each dgrad epilogue.
"""
import torch

from . import kernels as K
from . import tape as _tape


_TAPES = {}                  # id(first weight_v Parameter of a stack) -> (call tapes, leases); removed when the Parameter dies


def _stack_state(v0):
    """(call-tape cache, forward leases) of one sub-discriminator stack.  Kept OUTSIDE the Parameter (a Region holds PyCapsules:
    in the Parameter's __dict__ they made torch.save(model) / pickling fail after the first training step); keyed by id() with a
    finalizer -- a tensor cannot key a WeakKeyDictionary (its == is element-wise)."""
    st = _TAPES.get(id(v0))
    if st is None:
        st = _TAPES[id(v0)] = ({}, {})
        __import__("weakref").finalize(v0, _TAPES.pop, id(v0), None)
    return st


def _stack_tapes(v0):
    return _stack_state(v0)[0]


class _Lease:
    """A taped forward's activation buffers are the tape's: a second forward with the same key would replay into them while the
    first one's backward still needs them (two forwards of one stack before either backward: OSP_SHARE_REAL, any double forward).
    The forward therefore takes a lease on (key, slot) -- the slot is part of the tape key, i.e. a second buffer set is recorded for
    an overlapping forward -- and gives it back when its backward has been issued or its graph is dropped."""

    def __init__(self, table, key):
        self.table, self.key = table, key
        table[key] = True

    def release(self):
        if self.table is not None:
            self.table.pop(self.key, None)
            self.table = None

    __del__ = release


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


def conv2d_fwd(x, w, bias, KH, KW, sh, sw, ph, pw, slope, out_bf16):
    """x (U,H,W,C) channels-last; w native (Cout, KH, KW, Cin) -> (U,Ho,Wo,Cout), optional fused LeakyReLU."""
    U, H, W, C = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    out = torch.empty((U, Ho, Wo, Cout), device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    K.conv2d_gemm_bf16(x.view(U * H * W, C), w, Cout, M=U * Ho * Wo, Trows=Ho * Wo, Wrows=Wo, Hin=H, Win=W, cin=C,
                       taps=KH * KW, KW=KW, a_step_h=sh, a_tapstep_h=1, a_off_h=-ph, a_step=sw, a_tapstep=1, a_off=-pw,
                       w_strides=(KH * KW * C, KW * C, C, 1), out=out.view(U * Ho * Wo, Cout), ldc=Cout, Tc=Ho * Wo, Wc=Wo,
                       epi=K.EPI_LRELU if slope is not None else K.EPI_NONE, bias=bias, slope=slope or 0.0)
    return out


def transpose_weight2d(w):
    """native (Cout, KH, KW, Cin) -> dgrad layout (Cin, KH, KW, Cout), bf16."""
    t = w.permute(3, 1, 2, 0).contiguous()
    return t if t.dtype == torch.bfloat16 else K.cast_bf16(t)


def conv2d_dgrad(dy, wt, H, W, KH, KW, sh, sw, ph, pw, *, lrelu_y=None, extra=None, slope=0.1, out_bf16=False, out=None):
    """dx (U,H,W,Cin) of a strided conv2d.  Each output phase (h % sh, w % sw) only sees a sub-sampled kernel; all phases
    run in ONE launch (osp_conv2d_dgrad_bf16, csrc/gemm_bf16.hip), with the LeakyReLU backward of the previous layer and
    its feature-matching gradient (``extra``) fused into the epilogue.  ``out``: contiguous destination (U,H,W,Cin) or None."""
    U, Ho, Wo, Cout = dy.shape
    Cin = wt.shape[0]
    dx = out if out is not None else torch.empty((U, H, W, Cin), device=dy.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    isbf = lambda t: int(t is not None and t.dtype == torch.bfloat16)                 # noqa: E731
    K.call("osp_conv2d_dgrad_bf16", dy, isbf(dy), wt, isbf(wt), dx, isbf(dx), U, H, W, Ho, Wo, Cin, Cout, KH, KW, sh, sw, ph, pw,
           K.EPI_LRELU_BWD if lrelu_y is not None else K.EPI_NONE, lrelu_y, isbf(lrelu_y), extra, isbf(extra), float(slope))
    return dx


def conv2d_wgrad(dy, x, KH, KW, sh, sw, ph, pw):
    U, Ho, Wo, Cout = dy.shape
    _, H, W, Cin = x.shape
    dw = torch.zeros((Cout, KH, KW, Cin), device=dy.device, dtype=torch.float32)
    db = torch.zeros((Cout,), device=dy.device, dtype=torch.float32)
    K.conv2d_wgrad_bf16(dy.view(U * Ho * Wo, Cout), x.view(U * H * W, Cin), dw, db, M=U * Ho * Wo, Trows=Ho * Wo, Wrows=Wo,
                        Hin=H, Win=W, n=Cout, cin=Cin, taps=KH * KW, KW=KW, pad_h=ph, pad_w=pw, step_h=sh, step_w=sw)
    return dw, db


# =================================================================================================== generic stack
def wnorm_packed(v, g, want_f32, want_t):
    """osp_wnorm_fwd with a per-parameter cache: inside one training step the discriminator weights are packed once and
    reused by the real / generated passes of the generator phase and by the discriminator phase (the reference
    re-evaluates torch's weight_norm parametrisation on every call).  The pack lives on the ``weight_v`` Parameter object
    (so it dies with the module; a re-used allocation can never alias it) and is valid while neither the optimizer
    (values.param_epoch: the fused AdamW writes the arena through a raw pointer) nor a torch in-place op
    (Tensor._version) nor a re-binding of ``.data`` (data_ptr) touched v / g."""
    from . import values
    stamp = (values.param_epoch(), v._version, g._version, v.data_ptr(), g.data_ptr())
    hit = getattr(v, "_osp_wn_pack", None)
    if hit is not None and hit[0] == stamp and (hit[1][1] is not None or not want_f32) and (hit[1][2] is not None or not want_t):
        return hit[1]
    if hit is not None and hit[0] == stamp:                       # same weights, more outputs wanted: keep the union
        want_f32, want_t = want_f32 or hit[1][1] is not None, want_t or hit[1][2] is not None
    pack = K.wnorm_fwd_multi([(v.detach(), g.detach(), want_f32, want_t)], reuse=[hit[1] if hit is not None else None])[0]
    v._osp_wn_pack = (stamp, pack)
    return pack


def wnorm_pack_many(convs, want_f32):
    """Weight-norm packs of a list of conv modules (``weight_v`` / ``weight_g``) whose cached pack is stale, in ONE launch
    (osp_wnorm_fwd_multi) -- the per-conv osp_wnorm_fwd launches were 48 per discriminator pass.  ``want_f32``: also produce the
    f32 native copy for every conv (parity mode) or only for the narrow first layers (None).  Fills the same per-parameter cache
    wnorm_packed() reads; returns True when anything was launched (the caller must then order its side streams behind it)."""
    from . import values
    todo = []
    for conv in convs:
        v, g = conv.weight_v, conv.weight_g
        cout, cin, KH, KW = v.shape[0], v.shape[1], v.shape[3], v.shape[2]      # reference layout (Cout, Cin, k_w/freq, k_h/time)
        small = cin == 1 and cout in (16, 32, 64) and KH * KW <= cout
        f32 = bool(want_f32) or small
        stamp = (values.param_epoch(), v._version, g._version, v.data_ptr(), g.data_ptr())
        hit = getattr(v, "_osp_wn_pack", None)
        if hit is not None and hit[0] == stamp and (hit[1][1] is not None or not f32) and hit[1][2] is not None:
            continue
        todo.append((v, g, f32, stamp, hit[1] if hit is not None else None))
    if not todo:
        return False
    # the buffers of a stale pack are REWRITTEN (same addresses over the epochs: the stacks' call tapes hold them); ordering: this
    # launch is queued after everything that read the old contents (the caller joins the discriminator streams first)
    packs = K.wnorm_fwd_multi([(v.detach(), g.detach(), f32, True) for v, g, f32, _, _ in todo], reuse=[t[4] for t in todo])
    for (v, g, f32, stamp, _), pack in zip(todo, packs):
        v._osp_wn_pack = (stamp, pack)
    return True


def g_stream_is_side():
    """True when the current stream is not the device's default stream (a sub-discriminator stream)."""
    return torch.cuda.current_stream() != torch.cuda.default_stream()


class ConvStackFn(torch.autograd.Function):
    """A whole DiscriminatorP / DiscriminatorR conv stack with weight norm folded in.

    forward(x (U,H,W,1) f32, spec, slope, v0, g0, b0, ..., v5, g5, b5)  -- the *raw* weight_norm parameters (leaves).
    ``spec`` rows (KH, KW, sh, sw, ph, pw) in native (H, W) orientation; layers 0..4 get LeakyReLU, layer 5 is conv_post.
    Per layer one osp_wnorm_fwd packs g*v/||v|| into the bf16 native / transposed layouts; the backward turns the native
    f32 weight gradient into (dv, dg) with osp_wnorm_bwd and ACCUMULATES into the parameters' gradient-arena slots
    (so the Function returns None for parameters, like every other op of this package).
    outputs: y1..y5 (bf16, LeakyReLU applied) and the score map s (U,H5,W5,1) f32.
    """

    @staticmethod
    def forward(ctx, x, spec, slope, *params):
        # ``slope`` may be (slope, u0): the first u0 sequences are a no-grad branch (the real waves in the generator
        # phase) that shares the forward launches with the rest; the outputs then come as 6 no-grad tensors followed by
        # 6 differentiable ones (views of the same buffers) and the backward runs over the sequences from u0 on only.
        u0, holder = 0, None
        if isinstance(slope, tuple):
            holder = slope[2] if len(slope) > 2 else None
            slope, u0 = slope[0], slope[1]
        ctx.set_materialize_grads(False)
        vs, gs, bs = params[0::3], params[1::3], params[2::3]
        need_w = [bool(ctx.needs_input_grad[3 + 3 * i]) for i in range(6)]
        need_x = bool(ctx.needs_input_grad[0])
        packs = []
        x = x.contiguous()
        for i in range(6):
            KH, KW, sh, sw, ph, pw = spec[i]
            cout, cin = vs[i].shape[0], vs[i].shape[1]
            small = (cin == 1 and cout in (16, 32, 64) and KH * KW <= cout)
            # the transposed (dgrad) pack is always produced: the no-grad real pass of the generator phase comes first
            # and would otherwise force a second osp_wnorm_fwd for the generated pass
            packs.append(wnorm_packed(vs[i], gs[i], small, True))
        biases = [b.detach() for b in bs]

        def layers(xin):
            """The stack's six launches: a pure kernel sequence (recorded once per shape as a call tape, optispeech_amd/tape.py)."""
            out, h = [], xin
            for i in range(6):
                KH, KW, sh, sw, ph, pw = spec[i]
                cout, cin = vs[i].shape[0], vs[i].shape[1]
                wn, wn32, wt, inv = packs[i]
                lr = slope if i < 5 else None
                if wn32 is not None and cin == 1 and cout in (16, 32, 64) and KH * KW <= cout:
                    U, H, W = h.shape[0], h.shape[1], h.shape[2]
                    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
                    h = K.smallcin_fwd(h, wn32.view(cout, -1), biases[i], U=U, Hin=H, Win=W, Ho=Ho, Wo=Wo, cout=cout, KH=KH,
                                       KW=KW, sh=sh, sw=sw, ph=ph, pw=pw, slope=lr, out_bf16=i < 5).view(U, Ho, Wo, cout)
                else:
                    h = conv2d_fwd(h, wn, biases[i], KH, KW, sh, sw, ph, pw, lr, i < 5)
                out.append(h)
            return tuple(out)

        # (the generator-phase and the discriminator-phase pass of a step see the same shapes: ``any(need_w)`` keeps their
        # activation buffers apart, as the eager allocations were)
        key = ("fwd", tuple(x.shape), spec, float(slope), any(need_w), tuple(t.data_ptr() for pk in packs for t in pk if t is not None),
               tuple(b.data_ptr() for b in biases))
        tapes, leases = _stack_state(vs[0])
        slot = 0
        if need_x or any(need_w):
            while (key, slot) in leases:                         # an earlier forward with this key still waits for its backward
                slot += 1
            ctx.lease = _Lease(leases, (key, slot))
        acts = list(_tape.run(tapes, key + (slot,), [x], layers, "disc stack forward"))
        if need_x or any(need_w):
            ctx.save_for_backward(x[u0:], *[a[u0:] for a in acts[:5]],
                                  *[t for pk in packs for t in (pk[0], pk[2], pk[3]) if t is not None])
            ctx.pack_layout = [(pk[0] is not None, pk[2] is not None, pk[3] is not None) for pk in packs]
            ctx.params = params
            ctx.cfg = (spec, slope, need_x, need_w)
            ctx.u0, ctx.U = u0, x.shape[0]
        if holder is not None:
            # everything a later backward over the WHOLE batch needs (ConvStackReplayFn): the discriminator phase of the
            # same step sees the same waves and the same weights, so it replays this forward instead of repeating it
            from . import values
            holder["rec"] = (values.param_epoch(), x, list(acts), [(pk[0], pk[2], pk[3]) for pk in packs], spec, slope)
        if u0:
            head = tuple(a[:u0] for a in acts)
            ctx.mark_non_differentiable(*head)
            return head + tuple(a[u0:] for a in acts)
        return tuple(acts)

    @staticmethod
    def backward(ctx, *douts):
        d1, d2, d3, d4, d5, ds = douts[-6:]                      # (with a no-grad head the first 6 gradients are None)
        spec, slope, need_x, need_w = ctx.cfg
        saved = list(ctx.saved_tensors)
        x, acts = saved[0], saved[1:6]
        rest = saved[6:]
        packs = []
        for has in ctx.pack_layout:
            item = []
            for flag in has:
                item.append(rest.pop(0) if flag else None)
            packs.append(item)                                  # (wn, wt, inv)
        # the gradient of the WHOLE input (need_x with a no-grad head): the head rows belong to the real waves, which carry no
        # gradient -- zeros there (one memset keeps NaN / Inf of uninitialised memory away from anything that might reduce over
        # the batch dimension later); the last dgrad launch writes the rest in place
        g = _stack_backward(x, acts, packs, ctx.params, spec, slope, need_x, need_w, (d1, d2, d3, d4, d5), ds,
                            full_rows=ctx.U if (need_x and ctx.u0) else 0)
        ctx.lease.release()            # the backward's launches are queued behind the forward's on the stack's stream: the buffers are free
        return (g if need_x else None, None, None) + (None,) * len(ctx.params)


def _stack_backward(x, acts, packs, params, spec, slope, need_x, need_w, dfm, ds, full_rows=0):
    """Backward of a conv stack over (x, activations y1..y5): weight gradients into the arena (gsink), returns d x or None.
    ``dfm`` = gradients w.r.t. the feature maps y1..y5 (or None), ``ds`` = gradient w.r.t. the score map.  ``full_rows`` > 0: d x is
    returned for that many rows, the leading ``full_rows - x.shape[0]`` of them (a no-grad head of the forward) zero.

    The launches themselves are a pure kernel sequence (``launches`` below): recorded once per shape as a call tape and replayed
    (optispeech_amd/tape.py); what is not a launch -- the data-parallel ready signal, the stream join -- stays here."""
    from .ops import gsink
    vs, gs, bs = params[0::3], params[1::3], params[2::3]
    sizes = [vs[i].numel() if need_w[i] else 0 for i in range(6)]
    offs = [sum(sizes[:i]) for i in range(6)]
    sinks = [(gsink(vs[i]), gsink(gs[i]), gsink(bs[i])) if need_w[i] else None for i in range(6)]
    n_in = 6
    ins = [x] + list(acts) + [None if ds is None else ds.contiguous()] + [d.contiguous() if d is not None else None for d in dfm]
    for k, t in enumerate(ins):                                  # (the tape patches f32 / bf16 buffers by address: dtype is part of the key)
        if t is not None and k >= n_in and t.dtype not in (torch.float32, torch.bfloat16):
            ins[k] = t.float()

    def launches(x, a1, a2, a3, a4, a5, ds, e1, e2, e3, e4, e5):
        acts, extras = (a1, a2, a3, a4, a5), (e1, e2, e3, e4, e5)
        if ds is None:
            ds = torch.zeros((x.shape[0],) + tuple(acts[4].shape[1:3]) + (1,), device=x.device, dtype=torch.float32)
        g = ds
        # one zero-filled buffer for the native-layout weight gradients of all layers (one fill instead of six)
        flat = torch.zeros((sum(sizes),), device=g.device, dtype=torch.float32) if sum(sizes) else None
        wn_items = []                                             # (dW native f32, v, g, 1/||v||, dv, dg) of every layer: ONE launch
        for i in range(5, -1, -1):
            inp = acts[i - 1] if i > 0 else x
            KH, KW, sh, sw, ph, pw = spec[i]
            wn, wt, inv = packs[i]
            cout, cin = vs[i].shape[0], vs[i].shape[1]
            if need_w[i]:
                dw = flat[offs[i]:offs[i] + sizes[i]].view(cout, KH, KW, cin)
                dv, dg, db = sinks[i]
                if cin == 1 and cout in (16, 32, 64) and KH * KW <= cout:
                    K.smallcin_wgrad(inp, g, dw, db, U=inp.shape[0], Hin=inp.shape[1], Win=inp.shape[2], Ho=g.shape[1],
                                     Wo=g.shape[2], cout=cout, KH=KH, KW=KW, sh=sh, sw=sw, ph=ph, pw=pw)
                else:
                    U, Ho, Wo = g.shape[0], g.shape[1], g.shape[2]
                    H, W = inp.shape[1], inp.shape[2]
                    K.conv2d_wgrad_bf16(g.view(U * Ho * Wo, cout), inp.view(U * H * W, cin), dw, db, M=U * Ho * Wo,
                                        Trows=Ho * Wo, Wrows=Wo, Hin=H, Win=W, n=cout, cin=cin, taps=KH * KW, KW=KW, pad_h=ph,
                                        pad_w=pw, step_h=sh, step_w=sw)
                wn_items.append((dw, vs[i].detach(), gs[i].detach(), inv, dv, dg))
            if i > 0 and (need_x or any(need_w[:i])):
                g = conv2d_dgrad(g, wt, inp.shape[1], inp.shape[2], KH, KW, sh, sw, ph, pw, lrelu_y=inp, extra=extras[i - 1],
                                 slope=slope, out_bf16=True)
            elif i == 0 and need_x:
                out = None
                if full_rows:
                    full = torch.empty((full_rows,) + tuple(inp.shape[1:]), device=g.device, dtype=torch.float32)
                    head = full_rows - inp.shape[0]
                    K.call("osp_memset", full, 0, head * inp[0].numel() * 4)
                    out = full[head:]
                g = conv2d_dgrad(g, wt, inp.shape[1], inp.shape[2], KH, KW, sh, sw, ph, pw, out_bf16=False, out=out)
                if full_rows:
                    g = full
            else:
                g = None
                break
        if wn_items:
            K.wnorm_bwd_multi(wn_items)
        return g

    key = ("bwd", tuple(x.shape), spec, float(slope), need_x, tuple(need_w), full_rows,
           tuple(None if t is None else (tuple(t.shape), t.dtype) for t in ins[n_in:]),
           tuple(t.data_ptr() for pk in packs for t in pk if t is not None),
           tuple(t.data_ptr() for sk in sinks if sk is not None for t in sk))
    g = _tape.run(_stack_tapes(vs[0]), key, ins, launches, "disc stack backward")
    if any(need_w):
        if all(need_w):
            from .dp import reduce_ready
            reduce_ready(list(params))                           # data parallel: this stack's gradient slice is complete -> all-reduce it now
        if g_stream_is_side():
            # Parameter gradients are written straight into the gradient arena (no AccumulateGrad node), so the autograd
            # engine does not know that the stream backward() was called from must wait for this node's stream: say so.
            side = torch.cuda.current_stream()
            torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream().wait_stream(side))
    return g


class ConvStackReplayFn(torch.autograd.Function):
    """Discriminator-phase view of a stack whose forward already ran in the generator phase of the same step (same waves,
    same weights -> the same activations): forward returns the recorded score map, backward is the stack's full backward
    (weight gradients for every layer) over the whole 2B batch.  ``rec`` comes from ConvStackFn.forward(holder=...)."""

    @staticmethod
    def forward(ctx, rec, *params):
        _, x, acts, packs, spec, slope = rec
        ctx.rec, ctx.params = rec, params
        ctx.need_w = [bool(ctx.needs_input_grad[1 + 3 * i]) for i in range(6)]
        return acts[5].detach().clone()

    @staticmethod
    def backward(ctx, ds):
        _, x, acts, packs, spec, slope = ctx.rec
        _stack_backward(x, acts[:5], packs, ctx.params, spec, slope, False, ctx.need_w, (None,) * 5, ds)
        return (None,) + (None,) * len(ctx.params)


# =================================================================================================== f32 parity mode
# The exact-parity mode keeps f32 activations and weights.  The MFMA conv kernels take bf16 operands, so every f32 operand
# is split into hi + lo bf16 halves (x = hi + lo with |lo| <= 2^-9 |x|) and a product becomes hi*hi + lo*hi + hi*lo
# (the dropped lo*lo term is 2^-18 relative): three launches of the same kernels with f32 accumulation give ~2e-5 relative
# accuracy -- two orders inside the 1e-3 parity tolerance -- without a separate f32 conv2d code path (and without MIOpen).
def _split_bf16(x32):
    hi = K.cast_bf16(x32)
    return hi, K.cast_bf16(x32 - hi.float())


def precise_conv2d_fwd(x32, w32, KH, KW, sh, sw, ph, pw):
    xh, xl = _split_bf16(x32.contiguous())
    wh, wl = _split_bf16(w32.contiguous())
    y = conv2d_fwd(xh, wh, None, KH, KW, sh, sw, ph, pw, None, False)
    y += conv2d_fwd(xl, wh, None, KH, KW, sh, sw, ph, pw, None, False)
    y += conv2d_fwd(xh, wl, None, KH, KW, sh, sw, ph, pw, None, False)
    return y


def precise_conv2d_dgrad(dy32, wt32, H, W, KH, KW, sh, sw, ph, pw):
    dh, dl = _split_bf16(dy32.contiguous())
    th, tl = _split_bf16(wt32.contiguous())
    dx = conv2d_dgrad(dh, th, H, W, KH, KW, sh, sw, ph, pw)
    dx += conv2d_dgrad(dl, th, H, W, KH, KW, sh, sw, ph, pw)
    dx += conv2d_dgrad(dh, tl, H, W, KH, KW, sh, sw, ph, pw)
    return dx


def precise_conv2d_wgrad(dy32, x32, KH, KW, sh, sw, ph, pw):
    dh, dl = _split_bf16(dy32.contiguous())
    xh, xl = _split_bf16(x32.contiguous())
    dw, db = conv2d_wgrad(dh, xh, KH, KW, sh, sw, ph, pw)
    dw2, db2 = conv2d_wgrad(dl, xh, KH, KW, sh, sw, ph, pw)
    dw3, _ = conv2d_wgrad(dh, xl, KH, KW, sh, sw, ph, pw)
    return dw + dw2 + dw3, db + db2


class ConvStackPreciseFn(torch.autograd.Function):
    """ConvStackFn for the f32 parity mode: same stack, f32 activations, split-bf16 products (see above); the Cin = 1 first
    layer runs on the exact-f32 VALU kernels of csrc/smallcin.hip.  Outputs y1..y5 (f32, LeakyReLU applied) and the score map."""

    @staticmethod
    def forward(ctx, x, spec, slope, *params):
        ctx.set_materialize_grads(False)
        vs, gs, bs = params[0::3], params[1::3], params[2::3]
        need_w = [bool(ctx.needs_input_grad[3 + 3 * i]) for i in range(6)]
        need_x = bool(ctx.needs_input_grad[0])
        packs, acts, h = [], [], x.contiguous()
        for i in range(6):
            KH, KW, sh, sw, ph, pw = spec[i]
            cout, cin = vs[i].shape[0], vs[i].shape[1]
            _, wn32, _, inv = wnorm_packed(vs[i], gs[i], True, True)
            packs.append((wn32, inv))
            if cin == 1:
                U, H, W = h.shape[0], h.shape[1], h.shape[2]
                Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
                h = K.smallcin_fwd(h, wn32.view(cout, -1), bs[i].detach(), U=U, Hin=H, Win=W, Ho=Ho, Wo=Wo, cout=cout, KH=KH,
                                   KW=KW, sh=sh, sw=sw, ph=ph, pw=pw, slope=slope if i < 5 else None,
                                   out_bf16=False).view(U, Ho, Wo, cout)
            else:
                h = precise_conv2d_fwd(h, wn32, KH, KW, sh, sw, ph, pw) + bs[i].detach()
                if i < 5:
                    h = torch.where(h > 0, h, h * slope)
            acts.append(h)
        if need_x or any(need_w):
            ctx.save_for_backward(x, *acts[:5], *[t for pk in packs for t in pk])
            ctx.params = params
            ctx.cfg = (spec, slope, need_x, need_w)
        return tuple(acts)

    @staticmethod
    def backward(ctx, d1, d2, d3, d4, d5, ds):
        from .ops import gsink
        spec, slope, need_x, need_w = ctx.cfg
        saved = list(ctx.saved_tensors)
        x, acts, rest = saved[0], saved[1:6], saved[6:]
        packs = [(rest[2 * i], rest[2 * i + 1]) for i in range(6)]
        vs, gs, bs = ctx.params[0::3], ctx.params[1::3], ctx.params[2::3]
        extras = (d1, d2, d3, d4, d5)
        if ds is None:
            ds = torch.zeros((x.shape[0],) + tuple(acts[4].shape[1:3]) + (1,), device=x.device, dtype=torch.float32)
        g = ds.contiguous().float()
        for i in range(5, -1, -1):
            inp = acts[i - 1] if i > 0 else x
            KH, KW, sh, sw, ph, pw = spec[i]
            wn32, inv = packs[i]
            cout, cin = vs[i].shape[0], vs[i].shape[1]
            if need_w[i]:
                if cin == 1:
                    dw = torch.zeros((cout, KH, KW, cin), device=g.device, dtype=torch.float32)
                    K.smallcin_wgrad(inp, g, dw, gsink(bs[i]), U=inp.shape[0], Hin=inp.shape[1], Win=inp.shape[2], Ho=g.shape[1],
                                     Wo=g.shape[2], cout=cout, KH=KH, KW=KW, sh=sh, sw=sw, ph=ph, pw=pw)
                else:
                    dw, db = precise_conv2d_wgrad(g, inp, KH, KW, sh, sw, ph, pw)
                    gsink(bs[i]).add_(db)
                K.wnorm_bwd(dw, vs[i].detach(), gs[i].detach(), inv, gsink(vs[i]), gsink(gs[i]))
            if (i > 0 and (need_x or any(need_w[:i]))) or (i == 0 and need_x):
                g = precise_conv2d_dgrad(g, wn32.permute(3, 1, 2, 0).contiguous(), inp.shape[1], inp.shape[2], KH, KW, sh, sw, ph, pw)
                if i > 0:                                        # LeakyReLU backward of the layer below (+ its fmap gradient)
                    if extras[i - 1] is not None:
                        g = g + extras[i - 1]
                    g = torch.where(inp > 0, g, g * slope)
            else:
                g = None
                break
        return (g if need_x else None, None, None) + (None,) * len(ctx.params)


MPD_SPEC = ((1, 5, 1, 3, 0, 2),) * 4 + ((1, 5, 1, 1, 0, 2), (1, 3, 1, 1, 0, 1))
# DiscriminatorR: (KH = k_time, KW = k_freq, sh, sw, ph, pw) per layer in native (frames, bins) orientation
MRD_SPEC = ((5, 7, 2, 2, 2, 3), (3, 5, 1, 2, 1, 2), (3, 5, 2, 2, 1, 2), (3, 3, 1, 2, 1, 1), (3, 3, 2, 2, 1, 1), (3, 3, 1, 1, 1, 1))


class L1MeanFn(torch.autograd.Function):
    """mean |target - y| with the gradient flowing to ``y`` only (FeatureMatchingLoss terms, disc/loss.py:71-85)."""

    @staticmethod
    def forward(ctx, target, y):
        out = torch.zeros((), device=y.device, dtype=torch.float32)
        target, y = target.contiguous(), y.contiguous()
        K.l1_sum(target, y, 1.0 / y.numel(), out)
        ctx.save_for_backward(target, y)
        return out

    @staticmethod
    def backward(ctx, gout):
        target, y = ctx.saved_tensors
        return None, K.l1_sign(target, y, 1.0 / y.numel(), gout.reshape(1).float().contiguous())


class FeatureMatchSumFn(torch.autograd.Function):
    """sum_i mean |target_i - y_i| over a whole list of feature-map pairs in ONE autograd node (FeatureMatchingLoss,
    disc/loss.py:71-85) and ONE launch per 32 pairs (osp_l1_sum_multi: the |a-b| reductions of all pairs accumulate into one
    scalar); the backward writes sign(y - target) / numel per pair, again one launch.  Inputs: n targets, then n generated maps."""

    @staticmethod
    def forward(ctx, n, *maps):
        tg, ys = [t.contiguous() for t in maps[:n]], [t.contiguous() for t in maps[n:]]
        out = torch.zeros((), device=ys[0].device, dtype=torch.float32)
        K.l1_sum_multi(tg, ys, out)
        ctx.save_for_backward(*tg, *ys)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, gout):
        n, saved = ctx.n, ctx.saved_tensors
        g = gout.reshape(1).float().contiguous()
        grads = K.l1_sign_multi(list(saved[:n]), list(saved[n:]), g)
        return (None,) + (None,) * n + tuple(grads)


class PeriodFoldFn(torch.autograd.Function):
    """wave (B, T) -> (B * period, 1, ceil(T / period), 1) period-column sequences incl. the right reflect pad
    (DiscriminatorP.forward, _discriminators.py:63-72): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, x, period):
        ctx.period, ctx.T = period, x.shape[1]
        y = K.period_fold(x.contiguous(), period)
        return y.view(y.shape[0], 1, y.shape[1], 1)

    @staticmethod
    def backward(ctx, dy):
        return K.period_fold(dy.reshape(dy.shape[0], -1).contiguous().float(), ctx.period, backward=True, T=ctx.T), None


class SplitHalvesFn(torch.autograd.Function):
    """(o[:B], o[B:]) of a score map computed on the concatenated (real, generated) batch.  Plain slicing costs two zero fills,
    two copies and an add per map in the backward (SliceBackward x 2 + accumulation); here the backward is one cat."""

    @staticmethod
    def forward(ctx, o, B):
        ctx.B, ctx.shape = B, o.shape
        return o[:B], o[B:]

    @staticmethod
    def backward(ctx, gr, gg):
        B, shape = ctx.B, ctx.shape
        if gr is None:
            gr = gg.new_zeros((B,) + tuple(shape[1:]))
        if gg is None:
            gg = gr.new_zeros((shape[0] - B,) + tuple(shape[1:]))
        return torch.cat([gr, gg], 0), None


class HingeSumFn(torch.autograd.Function):
    """sum_i mean(clamp(1 + sgn_i * x_i, min=0)) over a list of score maps in one node and one launch (GeneratorLoss /
    DiscriminatorLoss, disc/loss.py:16-65).  ``sgns``: tuple of +-1 per tensor."""

    @staticmethod
    def forward(ctx, sgns, *xs):
        xs = [x.contiguous().float() for x in xs]
        out = torch.zeros((), device=xs[0].device, dtype=torch.float32)
        K.hinge_sum_multi(xs, sgns, out)
        ctx.save_for_backward(*xs)
        ctx.sgns = sgns
        return out

    @staticmethod
    def backward(ctx, gout):
        g = gout.reshape(1).float().contiguous()
        need = ctx.needs_input_grad[1:]
        idx = [i for i, nd in enumerate(need) if nd]
        grads = [None] * len(need)
        if idx:
            dxs = K.hinge_grad_multi([ctx.saved_tensors[i] for i in idx], [ctx.sgns[i] for i in idx], g)
            for i, dx in zip(idx, dxs):
                grads[i] = dx
        return (None,) + tuple(grads)
