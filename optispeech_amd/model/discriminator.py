"""Host-side mirror of optispeech/model/vocoder/wavenext/disc/{__init__,_discriminators,loss}.py.

VocosDiscriminator = MPD (5 periods) + MRD (3 resolutions) + hinge / feature-matching / mel / MR-STFT losses.
Every Conv2d stack runs on the hand-written kernels (optispeech_amd/disc_ops.py: bf16 MFMA conv-GEMMs in performance mode,
split-bf16 products of the same kernels in the f32 parity mode); the STFT magnitudes and the spectral losses are HIP kernels
too (csrc/stft.hip).  There is no torch conv2d / MIOpen path in the product: the torch reference the f32 mode is compared
against lives in tests/_torch_disc_ref.py.  Parameter names follow the reference's weight_norm schema (`weight_g` / `weight_v`).
"""
from types import SimpleNamespace

import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import precision, spectral
from ..disc_ops import (MPD_SPEC, MRD_SPEC, ConvStackFn, ConvStackPreciseFn, ConvStackReplayFn, FeatureMatchSumFn, HingeSumFn,
                        L1MeanFn, PeriodFoldFn, SplitHalvesFn, wnorm_pack_many)


class BaseVocoderDiscriminator(nn.Module):
    """optispeech/model/discriminator/__init__.py:11-23."""

    def forward_disc(self, wav, wav_hat):
        raise NotImplementedError

    def forward_gen(self, wav, wav_hat):
        raise NotImplementedError

    def forward_val(self, wav, wav_hat):
        raise NotImplementedError


class _WNConv2d(nn.Module):
    """weight_norm(Conv2d): w = g * v / ||v||  (norm over all dims but 0), keys bias / weight_g / weight_v."""

    def __init__(self, cin, cout, kernel, stride=(1, 1), padding=(0, 0)):
        super().__init__()
        conv = nn.Conv2d(cin, cout, kernel, stride, padding=padding)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(conv.bias.detach().clone())
        v = conv.weight.detach().clone()
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, 1, 1, 1))
        self.weight_v = nn.Parameter(v)


class DiscriminatorP(nn.Module):
    """_discriminators.py:41-97."""

    def __init__(self, period: int, kernel_size: int = 5, stride: int = 3, lrelu_slope: float = 0.1):
        super().__init__()
        self.period, self.lrelu_slope = period, lrelu_slope
        p = (kernel_size // 2, 0)
        ch = [1, 32, 128, 512, 1024]
        self.convs = nn.ModuleList([_WNConv2d(ch[i], ch[i + 1], (kernel_size, 1), (stride, 1), p) for i in range(4)]
                                   + [_WNConv2d(1024, 1024, (kernel_size, 1), (1, 1), p)])
        self.conv_post = _WNConv2d(1024, 1, (3, 1), (1, 1), (1, 0))

    def forward(self, x, nograd_head=0):
        """nograd_head = B0 (bf16 mode only): the first B0 waves are a no-grad branch sharing the launches; returns
        ((score, fmap) of the head, (score, fmap) of the rest)."""
        b = x.shape[0]
        if precision.is_bf16() and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and _PERIOD_FOLD:
            # reflect pad and period folding in one launch (csrc/fused_small.hip: period_fold_kernel)
            seq = PeriodFoldFn.apply(x, self.period)
        else:
            x = x.unsqueeze(1)
            b, c, t = x.shape
            if t % self.period != 0:
                n_pad = self.period - (t % self.period)
                x = F.pad(x, (0, n_pad), "reflect")
                t = t + n_pad
            seq = x.view(b, t // self.period, self.period).transpose(1, 2).reshape(b * self.period, 1, t // self.period, 1).contiguous()
        # channels-last sequences (B*period, T/period, 1): every period column is an independent 1-D signal
        args = []
        for conv in list(self.convs) + [self.conv_post]:
            args += [conv.weight_v, conv.weight_g, conv.bias]
        if precision.is_bf16():
            if nograd_head:
                self._holder = {}
                o = ConvStackFn.apply(seq, MPD_SPEC, (self.lrelu_slope, nograd_head * self.period, self._holder), *args)
                (_, r2, r3, r4, r5, rs), (_, y2, y3, y4, y5, s) = o[:6], o[6:]
                return (rs.reshape(nograd_head, -1), [r2, r3, r4, r5, rs]), (s.reshape(b - nograd_head, -1), [y2, y3, y4, y5, s])
            y1, y2, y3, y4, y5, s = ConvStackFn.apply(seq, MPD_SPEC, self.lrelu_slope, *args)
            return s.view(b, -1), [y2, y3, y4, y5, s]
        # f32 parity mode on the same kernels (split-bf16 products, disc_ops.ConvStackPreciseFn); feature maps come out
        # channels-last, which the (layout-agnostic) mean losses do not care about
        y1, y2, y3, y4, y5, s = ConvStackPreciseFn.apply(seq, MPD_SPEC, self.lrelu_slope, *args)
        return s.view(b, -1), [y2, y3, y4, y5, s]


class DiscriminatorR(nn.Module):
    """_discriminators.py:139-216; the rectangular-window magnitude spectrogram comes from the HIP STFT."""

    def __init__(self, resolution, channels: int = 64, lrelu_slope: float = 0.1):
        super().__init__()
        self.resolution, self.lrelu_slope = resolution, lrelu_slope
        spec = [((7, 5), (2, 2), (3, 2)), ((5, 3), (2, 1), (2, 1)), ((5, 3), (2, 2), (2, 1)), ((3, 3), (2, 1), (1, 1)),
                ((3, 3), (2, 2), (1, 1))]
        self.convs = nn.ModuleList([_WNConv2d(1 if i == 0 else channels, channels, k, s, p)
                                    for i, (k, s, p) in enumerate(spec)])
        self.conv_post = _WNConv2d(channels, 1, (3, 3), (1, 1), (1, 1))

    def spectrogram(self, x):
        n_fft, hop, win = self.resolution
        return spectral.stft_magnitude(x, n_fft, hop, None, None).transpose(1, 2)      # (B, freq, frames)

    def forward(self, x, nograd_head=0):
        if precision.is_bf16():
            n_fft, hop, win = self.resolution
            spec = spectral.stft_magnitude(x, n_fft, hop, None, None)              # (B, frames, bins), channels-last
            args = []
            for conv in list(self.convs) + [self.conv_post]:
                # reference weight (Cout, Cin, k_freq, k_time) is packed to native (Cout, KH = k_time, KW = k_freq, Cin)
                args += [conv.weight_v, conv.weight_g, conv.bias]
            if nograd_head:
                self._holder = {}
                o = ConvStackFn.apply(spec.unsqueeze(-1), MRD_SPEC, (self.lrelu_slope, nograd_head, self._holder), *args)
                r, y = o[:6], o[6:]
                return (r[5].reshape(nograd_head, -1), list(r)), (y[5].reshape(y[5].shape[0], -1), list(y))
            y1, y2, y3, y4, y5, s = ConvStackFn.apply(spec.unsqueeze(-1), MRD_SPEC, self.lrelu_slope, *args)
            return s.reshape(s.shape[0], -1), [y1, y2, y3, y4, y5, s]
        n_fft, hop, win = self.resolution
        spec = spectral.stft_magnitude(x, n_fft, hop, None, None)
        args = []
        for conv in list(self.convs) + [self.conv_post]:
            args += [conv.weight_v, conv.weight_g, conv.bias]
        ys = ConvStackPreciseFn.apply(spec.unsqueeze(-1), MRD_SPEC, self.lrelu_slope, *args)
        return ys[5].reshape(ys[5].shape[0], -1), list(ys)


def _replay_scores(d, B):
    """(real scores, generated scores) of sub-discriminator ``d`` from the forward its generator-phase call recorded, as a
    graph node whose backward yields the weight gradients (disc_ops.ConvStackReplayFn); None if nothing valid is recorded."""
    from .. import values
    rec = getattr(d, "_holder", {}).pop("rec", None)
    if rec is None or rec[0] != values.param_epoch():
        return None
    args = []
    for conv in list(d.convs) + [d.conv_post]:
        args += [conv.weight_v, conv.weight_g, conv.bias]
    s = ConvStackReplayFn.apply(rec, *args)
    return SplitHalvesFn.apply(s.reshape(2 * B, -1), B)


class _Multi(nn.Module):
    def _plist(self):
        """Flat parameter list, taken once (walking the module tree for `self.parameters()` four times a step was 0.3 ms of host
        time) and checked against the live tensors' ids on every use (a list cached forever goes stale when parameters
        are replaced -- remove_weight_norm, an overwriting _apply -- and `real_needs_grad` would then read dead tensors).  The
        check is over the cached (module, name) slots, not a tree walk."""
        hit = self.__dict__.get("_param_slots")
        if hit is not None:
            slots, plist = hit
            if all(m._parameters.get(n) is q for (m, n), q in zip(slots, plist)) and self.__dict__.get("_param_count") == sum(len(m._parameters) for m in self.__dict__["_param_mods"]):
                return plist
        slots, plist, mods = [], [], []
        for m in self.modules():
            mods.append(m)
            for n, q in m._parameters.items():
                if q is not None:
                    slots.append((m, n)); plist.append(q)
        object.__setattr__(self, "_param_slots", (slots, plist))
        object.__setattr__(self, "_param_mods", mods)
        object.__setattr__(self, "_param_count", sum(len(m._parameters) for m in mods))
        return plist

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ("_param_slots", "_param_mods", "_param_count"):         # a cache, not state: never copied by deepcopy / pickle
            st.pop(k, None)
        return st

    def forward_real(self, y):
        """Real-wave branch only -> (scores, feature maps) per sub-discriminator, under the caller's grad mode.  Inside
        one training step the discriminator weights and the real waves are the same in the generator phase and in the
        discriminator phase (the D optimizer steps after both), so OptiSpeech.training_step evaluates this branch once,
        WITH the autograd graph, and hands the result to both phases (the reference evaluates it twice)."""
        rs, frs = [], []
        for d in self.discriminators:
            r, fr = d(y)
            rs.append(r); frs.append(fr)
        return rs, frs

    def forward(self, y, y_hat, real=None, defer_join=False, pre=None, replay=False):
        """Real and generated waves go through separately: in the generator phase the real branch needs no
        backward at all (it only feeds the feature-matching targets), so it runs under no_grad -- unless ``real``
        (a forward_real result) is supplied, in which case only the generated branch runs."""
        rs, gs, frs, fgs = [], [], [], []
        if real is not None:
            rs, frs = real
            for d in self.discriminators:
                g, fg = d(y_hat)
                gs.append(g); fgs.append(fg)
            return rs, gs, frs, fgs
        # (a flat list, taken once: walking the module tree for `self.parameters()` four times a step was 0.3 ms of host time)
        real_needs_grad = any(p.requires_grad for p in self._plist())
        B = y.shape[0]
        # weight-norm packs of every conv of this family in one launch (cached per optimiser epoch: the generator phase, its
        # real / generated halves and the discriminator phase of a step share them)
        packed = wnorm_pack_many([c for d in self.discriminators for c in list(d.convs) + [d.conv_post]], not precision.is_bf16())
        if precision.is_bf16() and _DISC_STREAMS and y.is_cuda:
            ready = pre[1] if (pre is not None and not packed) else None      # packs just launched: the side streams wait for them too
            return self._forward_concurrent(pre[0] if pre is not None else torch.cat([y, y_hat], 0), B, real_needs_grad, defer_join,
                                            ready, replay)
        for d in self.discriminators:
            if real_needs_grad:                              # discriminator phase: one batch of 2B waves per launch
                o, fm = d(torch.cat([y, y_hat], 0))
                r, g = SplitHalvesFn.apply(o, B)
                fr, fg = [f[: f.shape[0] // 2] for f in fm], [f[f.shape[0] // 2:] for f in fm]
            elif precision.is_bf16():
                # generator phase: real (no gradient) and generated waves share every forward launch; the backward of the
                # stack only covers the generated half
                (r, fr), (g, fg) = d(torch.cat([y, y_hat], 0), nograd_head=B)
            else:
                with torch.no_grad():
                    r, fr = d(y)
                g, fg = d(y_hat)
            rs.append(r); gs.append(g); frs.append(fr); fgs.append(fg)
        return rs, gs, frs, fgs


    def _forward_concurrent(self, x, B, with_param_grads, defer_join=False, ready=None, replay=False):
        """The sub-discriminators are independent (own weights, own spectrogram / period folding): each one runs on its own
        HIP stream, so the many small launches of one (first / last layers, weight-norm packing, the narrow MRD layers)
        overlap the large GEMMs of another.  autograd replays every node's backward on the stream its forward ran on and
        joins the streams at the end of backward(), so the backward overlaps the same way."""
        main = torch.cuda.current_stream()
        fam = "p" if isinstance(self.discriminators[0], DiscriminatorP) else "r"
        streams = _disc_streams(id(self), len(self.discriminators), x.device, names=[fam + str(i) for i in range(len(self.discriminators))])
        assert len(streams) == len(self.discriminators)
        if ready is None:
            ready = main.record_event()
        outs, fms, hgs = [], [], []
        for d, st in zip(self.discriminators, streams):
            st.wait_event(ready)
            with torch.cuda.stream(st):
                rep = _replay_scores(d, B) if (with_param_grads and replay) else None
                if rep is not None:                          # discriminator phase on the generator phase's forward
                    out = (rep[0], rep[1], [], [])
                elif with_param_grads:                       # discriminator phase: one batch of 2B waves per launch
                    o, fm = d(x)
                    r_, g_ = SplitHalvesFn.apply(o, B)
                    out = (r_, g_, [f[: f.shape[0] // 2] for f in fm], [f[f.shape[0] // 2:] for f in fm])
                    inl = _INLINE["on"]
                    if inl is not None and _fused_losses(r_):
                        # (OSP_INLINE_D=1) this stack's hinge term as its own node on its own stream: its backward needs nothing of the OTHER
                        # stacks, so the stack goes from forward to backward without the phase-wide join in between
                        hgs.append(HingeSumFn.apply((-1.0, 1.0), r_, g_))
                        if True:
                            # ... and RUNS that backward right here (discriminator phase of the pipelined step): the phase's loss is
                            # sum_i c_i * hinge_i with host-known c_i, so stack i's gradient seed is a constant -- no loss node on the
                            # phase's calling stream whose kernels (and the join in front of them) every stack's backward would wait for
                            st.wait_event(inl["zeroed"])                   # the gradient arena was cleared on the phase's stream
                            with torch.autograd.set_multithreading_enabled(False):
                                torch.autograd.backward([hgs[-1]], [_const_scalar(inl["coef"][id(self)], x.device)])
                            hgs[-1] = hgs[-1].detach()
                else:                                        # generator phase: no-grad head = the real waves
                    (r, fr), (g, fg) = d(x, nograd_head=B)
                    out = (r, g, fr, fg)
                    if _FM_PER_STACK and _fused_losses(fg[0]):
                        # this stack's share of the feature-matching sum, on the stack's own stream: the |a - b| reduction runs beside the
                        # other stacks' GEMMs instead of after the join of all eight, and its backward (sign kernel) at the head of this
                        # stack's backward, again on this stream (autograd replays a node where its forward ran)
                        fms.append(FeatureMatchSumFn.apply(len(fg), *[a.detach() for a in fr], *fg))
            x.record_stream(st)
            outs.append(out)
        self._fm_partials = fms if len(fms) == len(self.discriminators) else None
        self._hinge_partials = hgs if len(hgs) == len(self.discriminators) else None
        rs, gs, frs, fgs = [], [], [], []
        for i, ((r, g, fr, fg), st) in enumerate(zip(outs, streams)):
            _PENDING.append((st, [r, g] + list(fr) + list(fg) + ([fms[i]] if self._fm_partials is not None else [])
                             + ([hgs[i]] if self._hinge_partials is not None else [])))
            rs.append(r); gs.append(g); frs.append(fr); fgs.append(fg)
        if not defer_join:
            join_streams()
        return rs, gs, frs, fgs


_DISC_STREAMS = os.environ.get("OSP_DISC_STREAMS", "1") != "0"
#: generator phase: the feature-matching loss as one node PER sub-discriminator on that stack's stream (round 5) instead of one node
#: per family after all stacks have joined
_FM_PER_STACK = os.environ.get("OSP_FM_PER_STACK", "1") != "0"
_PERIOD_FOLD = os.environ.get("OSP_PERIOD_FOLD", "1") != "0"
#: hinge / feature-matching means as one autograd node per loss term (fused reductions) instead of ~10 torch ops per map
_FUSED_LOSSES = os.environ.get("OSP_FUSED_LOSSES", "1") != "0"
_STREAMS = {}
#: HIP priority of the sub-discriminator streams (lower number = higher priority; 0 = default).  See OptiSpeech.__init__.
_DISC_PRIORITY = int(os.environ.get("OSP_PRIO_DISC", "0"))
_MAX_STREAMS = int(os.environ.get("OSP_DISC_MAX_STREAMS", "8"))     # streams per discriminator family
_PENDING = []
#: set by OptiSpeech._stage_d around forward_disc: {"coef": {id(family): d loss / d hinge_i}, "zeroed": event after zero_grad}
_INLINE = {"on": None}
_CONSTS = {}


def _const_scalar(v, device):
    """A persistent 0-d f32 device tensor holding ``v`` (gradient seeds of the inline per-stack backward)."""
    k = (float(v), str(device))
    t = _CONSTS.get(k)
    if t is None:
        if len(_CONSTS) > 64:
            _CONSTS.clear()
        t = _CONSTS[k] = torch.full((), float(v), device=device, dtype=torch.float32)
        torch.cuda.current_stream().synchronize()            # created once: every stream may read it from now on
    return t


def join_streams():
    """The current stream waits for every side stream with work queued since the last join; the tensors produced there are
    marked as used on the current stream (caching-allocator lifetime).  No host synchronisation."""
    main = torch.cuda.current_stream()
    while _PENDING:
        st, tensors = _PENDING.pop()
        main.wait_stream(st)
        for t in tensors:
            t.record_stream(main)


def on_side_stream(key, fn, inputs):
    """Run ``fn()`` on a persistent side stream (ordered after the current stream's work so far); joined by join_streams()."""
    dev = inputs[0].device
    st = _disc_streams(key, 1, dev, names=["spec"])[0]
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        out = fn()
    for t in inputs:
        t.record_stream(st)
    _PENDING.append((st, [t for t in (out if isinstance(out, (tuple, list)) else [out]) if isinstance(t, torch.Tensor)]))
    return out


_SKEWED = set()


def _skew(device, var):
    """Experiment knob: burn ``$var`` streams of torch's pool before the next one is taken (shifts which hardware queue the streams
    created afterwards land on: the runtime deals its hardware queues out round-robin)."""
    n = int(os.environ.get(var, "0"))
    if n and var not in _SKEWED:
        _SKEWED.add(var)
        _SKEWED.update(torch.cuda.Stream(device=device) for _ in range(n))


def _disc_streams(key, n, device, names=None):
    """n persistent side streams for ``key``.  The COUNT is part of the cache key: callers key by id(module), CPython re-uses the id
    of a collected object, and a five-stack family that inherited a dead three-stack family's entry was zipped against three
    streams -- two sub-discriminators silently dropped (found in round 3 by a test that builds both families in one process)."""
    k = (key, n, torch.device(device).index)
    if k not in _STREAMS:
        _skew(device, "OSP_SKEW_DISC")
        from .. import lanes
        if names is not None and lanes.managed():
            pool = [lanes.stream(nm, device) for nm in names[:min(n, _MAX_STREAMS)]]
        else:
            pool = [torch.cuda.Stream(device=device, priority=_DISC_PRIORITY) for _ in range(min(n, _MAX_STREAMS))]
        _STREAMS[k] = [pool[i % len(pool)] for i in range(n)]
    return _STREAMS[k]


class MultiPeriodDiscriminator(_Multi):
    """_discriminators.py:10-38."""

    def __init__(self, periods=(2, 3, 5, 7, 11)):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorP(period=p) for p in periods])


class MultiResolutionDiscriminator(_Multi):
    """_discriminators.py:100-136."""

    def __init__(self, resolutions=((1024, 256, 1024), (2048, 512, 2048), (512, 128, 512))):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorR(resolution=r) for r in resolutions])


def _fused_losses(t):
    return precision.is_bf16() and t.is_cuda and _FUSED_LOSSES


def _hinge_g(outs, raw=False):                                 # GeneratorLoss, disc/loss.py:16-32
    """raw=True (fused mode only): the un-normalised sum; the caller folds 1 / len(outs) into its weighted sum."""
    if _fused_losses(outs[0]):
        s = HingeSumFn.apply((-1.0,) * len(outs), *outs)
        return s if raw else s / len(outs)
    return sum(torch.mean(torch.clamp(1 - o, min=0)) for o in outs) / len(outs)


def _hinge_d(real, fake, raw=False):                           # DiscriminatorLoss, disc/loss.py:40-65
    if _fused_losses(real[0]):
        s = HingeSumFn.apply((-1.0,) * len(real) + (1.0,) * len(fake), *real, *fake)
        return s if raw else s / len(real)
    return sum(torch.mean(torch.clamp(1 - r, min=0)) + torch.mean(torch.clamp(1 + g, min=0))
               for r, g in zip(real, fake)) / len(real)


def _feature_matching(fr, fg, raw=False):                      # FeatureMatchingLoss, disc/loss.py:71-85
    if _fused_losses(fg[0][0]):
        tg = [a.detach() for dr in fr for a in dr]
        ys = [b for dg in fg for b in dg]
        s = FeatureMatchSumFn.apply(len(ys), *tg, *ys)
        return s if raw else s / len(fr)
    tot = 0
    for dr, dg in zip(fr, fg):
        for a, b in zip(dr, dg):
            if precision.is_bf16():
                tot = tot + L1MeanFn.apply(a.detach(), b)          # fused |a-b| mean; gradient to the generated branch
            else:
                tot = tot + torch.mean(torch.abs(a.detach() - b))   # target side carries no gradient (frozen D in the reference)
    return tot / len(fr)


class VocosDiscriminator(BaseVocoderDiscriminator):
    """disc/__init__.py:16-111.  Losses are returned as device scalars; log dicts hold device tensors (the
    reference calls .item() on each of them = one device sync apiece, base_lightning_module.py:132-148)."""

    def __init__(self, feature_extractor, loss_coeffs=None):
        super().__init__()
        self.feature_extractor = feature_extractor
        self.loss_coeffs = loss_coeffs or SimpleNamespace(lambda_mrd=1.0, lambda_mel=45.0, lambda_mr_stft=2.5)
        self.lambda_mel = self.loss_coeffs.lambda_mel
        self.lambda_mr_stft = self.loss_coeffs.lambda_mr_stft
        self.multiperioddisc = MultiPeriodDiscriminator()
        self.multiresddisc = MultiResolutionDiscriminator()
        fe = feature_extractor
        self.melspec_loss = spectral.MelSpecReconstructionLoss(fe.sample_rate, fe.n_fft, fe.hop_length, fe.win_length,
                                                               fe.n_feats, fe.f_min, fe.f_max)
        self.mr_stft_loss = spectral.MultiResolutionSTFTLoss()

    def forward_real(self, wav):
        with precision.disc_scope():
            return self.multiperioddisc.forward_real(wav), self.multiresddisc.forward_real(wav)

    def prepare_disc_inputs(self, wav, wav_hat):
        with precision.disc_scope():
            return self._prepare_disc_inputs(wav, wav_hat)

    def forward_disc(self, wav, wav_hat, real=None, pre=None, replay=False, inline_backward=None):
        """inline_backward = (1 / accumulation scale, event recorded after the gradient arena was cleared): every sub-discriminator
        runs its backward right behind its forward on its own stream (see _forward_concurrent); the returned loss is then a detached
        value -- there is nothing left to call backward() on."""
        with precision.disc_scope():
            inl = None
            if (inline_backward is not None and real is None and not replay and precision.is_bf16() and _DISC_STREAMS and _FM_PER_STACK
                    and wav.is_cuda and torch.is_grad_enabled()):
                n_mp, n_mr = len(self.multiperioddisc.discriminators), len(self.multiresddisc.discriminators)
                inl = {"coef": {id(self.multiperioddisc): inline_backward[0] / n_mp,
                                id(self.multiresddisc): inline_backward[0] * self.loss_coeffs.lambda_mrd / n_mr}, "zeroed": inline_backward[1]}
            _INLINE["on"] = inl
            try:
                return self._forward_disc(wav, wav_hat, real=real, pre=pre, replay=replay)
            finally:
                _INLINE["on"] = None

    def forward_gen(self, wav, wav_hat, real=None):
        with precision.disc_scope():
            return self._forward_gen(wav, wav_hat, real=real)

    def _prepare_disc_inputs(self, wav, wav_hat):
        """(concatenated waves, 'ready' event) for a later forward_disc: taken right after the generator forward so that the
        discriminator-phase forward does not have to wait for whatever the calling stream does in between (the generator's
        backward), only for the waves themselves."""
        if not (precision.is_bf16() and _DISC_STREAMS and wav.is_cuda):
            return None
        x = torch.cat([wav, wav_hat], 0)
        return x, torch.cuda.current_stream().record_event()

    def _forward_disc(self, wav, wav_hat, real=None, pre=None, replay=False):
        """replay=True: reuse the forward the generator phase of the SAME step ran on these waves (same weights -> same
        activations) and only run the backward with weight gradients; falls back to a fresh forward when nothing valid is recorded."""
        # both families are launched before either is joined: the eight stacks overlap across the family boundary too
        r_mp, g_mp, _, _ = self.multiperioddisc(y=wav, y_hat=wav_hat, real=real[0] if real is not None else None, defer_join=True, pre=pre, replay=replay)
        r_mr, g_mr, _, _ = self.multiresddisc(y=wav, y_hat=wav_hat, real=real[1] if real is not None else None, defer_join=True, pre=pre, replay=replay)
        join_streams()
        if _fused_losses(r_mp[0]):
            # the 1 / #sub-discriminators of each family and lambda_mrd are coefficients of ONE weighted-sum node
            from ..ops import weighted_sum
            hm, hr = getattr(self.multiperioddisc, "_hinge_partials", None), getattr(self.multiresddisc, "_hinge_partials", None)
            self.multiperioddisc._hinge_partials = self.multiresddisc._hinge_partials = None
            s_mp = weighted_sum(hm, [1.0] * len(hm)) if hm else _hinge_d(r_mp, g_mp, raw=True)
            s_mr = weighted_sum(hr, [1.0] * len(hr)) if hr else _hinge_d(r_mr, g_mr, raw=True)
            n_mp, n_mr = len(r_mp), len(r_mr)
            loss = weighted_sum([s_mp, s_mr], [1.0 / n_mp, self.loss_coeffs.lambda_mrd / n_mr])
            return loss, dict(loss_mp=s_mp.detach() / n_mp, loss_mrd=s_mr.detach() / n_mr)
        loss_mp, loss_mrd = _hinge_d(r_mp, g_mp), _hinge_d(r_mr, g_mr)
        loss = loss_mp + loss_mrd * self.loss_coeffs.lambda_mrd
        return loss, dict(loss_mp=loss_mp.detach(), loss_mrd=loss_mrd.detach())

    def _forward_gen(self, wav, wav_hat, real=None):
        _, g_mp, fr_mp, fg_mp = self.multiperioddisc(y=wav, y_hat=wav_hat, real=real[0] if real is not None else None, defer_join=True)
        _, g_mr, fr_mr, fg_mr = self.multiresddisc(y=wav, y_hat=wav_hat, real=real[1] if real is not None else None, defer_join=True)
        if precision.is_bf16() and _DISC_STREAMS and wav.is_cuda:
            # the spectral reconstruction losses do not depend on the discriminators: a ninth stream
            def spectral_losses():
                with precision.generator_scope():
                    return self._get_mel_loss(wav, wav_hat), self._get_mr_stft_loss(wav, wav_hat)
            mel_loss, mr_stft_loss = on_side_stream(("spectral", id(self)), spectral_losses, [wav, wav_hat])
        else:
            with precision.generator_scope():
                mel_loss = self._get_mel_loss(wav, wav_hat)
                mr_stft_loss = self._get_mr_stft_loss(wav, wav_hat)
        join_streams()
        lam = self.loss_coeffs.lambda_mrd
        if _fused_losses(g_mp[0]):
            from ..ops import weighted_sum
            n_mp, n_mr = len(g_mp), len(g_mr)
            pm, pr = getattr(self.multiperioddisc, "_fm_partials", None), getattr(self.multiresddisc, "_fm_partials", None)
            self.multiperioddisc._fm_partials = self.multiresddisc._fm_partials = None
            self.multiperioddisc._hinge_partials = self.multiresddisc._hinge_partials = None
            h_mp, h_mr = _hinge_g(g_mp, raw=True), _hinge_g(g_mr, raw=True)
            if pm and pr:
                # per-stack feature-matching terms (see _forward_concurrent) enter the ONE weighted-sum node directly; their family sums
                # are only needed for the log
                terms = [h_mp, h_mr] + list(pm) + list(pr) + [mel_loss, mr_stft_loss]
                coef = [1.0 / n_mp, lam / n_mr] + [1.0 / n_mp] * len(pm) + [lam / n_mr] * len(pr) + [1.0, 1.0]
                loss = weighted_sum(terms, coef)
                with torch.no_grad():
                    fm_mp = weighted_sum([t.detach() for t in pm], [1.0] * len(pm))
                    fm_mr = weighted_sum([t.detach() for t in pr], [1.0] * len(pr))
            else:
                fm_mp, fm_mr = _feature_matching(fr_mp, fg_mp, raw=True), _feature_matching(fr_mr, fg_mr, raw=True)
                loss = weighted_sum([h_mp, h_mr, fm_mp, fm_mr, mel_loss, mr_stft_loss], [1.0 / n_mp, lam / n_mr, 1.0 / n_mp, lam / n_mr, 1.0, 1.0])
            logs = dict(loss_gen_mp=h_mp.detach() / n_mp, loss_gen_mrd=h_mr.detach() / n_mr, loss_fm_mp=fm_mp.detach() / n_mp,
                        loss_fm_mrd=fm_mr.detach() / n_mr, mel_loss=mel_loss.detach(), mr_stft_loss=mr_stft_loss.detach())
            return loss, logs
        loss_gen_mp, loss_gen_mrd = _hinge_g(g_mp), _hinge_g(g_mr)
        loss_fm_mp, loss_fm_mrd = _feature_matching(fr_mp, fg_mp), _feature_matching(fr_mr, fg_mr)
        from ..ops import weighted_sum
        loss = weighted_sum([loss_gen_mp, loss_gen_mrd, loss_fm_mp, loss_fm_mrd, mel_loss, mr_stft_loss], [1.0, lam, 1.0, lam, 1.0, 1.0])
        logs = dict(loss_gen_mp=loss_gen_mp, loss_gen_mrd=loss_gen_mrd, loss_fm_mp=loss_fm_mp, loss_fm_mrd=loss_fm_mrd,
                    mel_loss=mel_loss, mr_stft_loss=mr_stft_loss)
        return loss, {k: v.detach() for k, v in logs.items()}

    def forward_val(self, wav, wav_hat):
        mel_loss = self._get_mel_loss(wav, wav_hat)
        mr_stft_loss = self._get_mr_stft_loss(wav, wav_hat)
        return mel_loss + mr_stft_loss, dict(mel_loss=mel_loss.detach(), mr_stft_loss=mr_stft_loss.detach())

    def _get_mel_loss(self, wav, wav_hat):
        return self.melspec_loss(wav_hat, wav) * self.lambda_mel

    def _get_mr_stft_loss(self, wav, wav_hat):
        sc, mag = self.mr_stft_loss(wav_hat, wav)
        return (sc + mag) * self.lambda_mr_stft
