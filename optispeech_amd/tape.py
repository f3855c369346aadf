"""Call tapes: a region of the training step recorded ONCE as a list of C-ABI calls and replayed from one C loop.

Why: the eager step needs 15-18 ms of host time to enqueue ~820 launches -- Python wrappers,
``torch.empty``, autograd bookkeeping, argument marshalling -- for ~17 ms of GPU time; hipGraph replay removes the host time but the
ROCm 7.2 runtime executes a captured multi-stream graph almost serially.  A tape keeps the eager schedule --
the same entry points, on the same streams, in the same order -- and removes the interpreter: ``_ospfast.tape_replay`` walks the
recorded calls in C (~1 us per call plus the launch itself).

What a region is: a Python callable that, given its input tensors, issues C-ABI calls (``_lib.call``), allocates with ``torch.empty``
and takes views -- nothing else.  While it is recorded

  * every C-ABI call is executed AND appended to the tape (entry index, marshalled arguments, stream);
  * pointers INTO the region's inputs are stored relative to the input and patched with the inputs' addresses at replay: inputs may
    live anywhere on the next step; every other tensor the calls saw is kept alive by the tape, so buffers allocated inside the region
    are persistent and weights / packs / gradient-arena slots keep their addresses by construction;
  * ATen operators are intercepted (``TorchDispatchMode``): allocation and view operators pass, the handful of element-wise / copy
    operators autograd and the host code still use are RE-ROUTED to entry points of csrc/ew.hip (so they land on the tape), anything
    else POISONS the recording -- the region then simply stays eager (and says so once): a kernel the tape does not know about can
    never be dropped silently.

Per-step scalars: dropout seeds live in device memory while tapes are in use (``rng.use_device_seed``; one ``osp_store_i64`` per step).

A replay returns fresh aliases of the region's persistent output buffers.  They are overwritten by the next replay of the same
region: callers are the autograd Functions of this package, which consume them within the step.
"""
import os
import warnings

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib

ENABLED = os.environ.get("OSP_TAPES", "1") != "0"
#: a taped Segment records a key at its N-th sighting (2: the first step with a batch signature runs eagerly, the second records,
#: later ones replay); OSP_TAPE_SEGMENT_AFTER=1 records at first sight (round 5's behaviour)
SEGMENT_RECORD_AFTER = int(os.environ.get("OSP_TAPE_SEGMENT_AFTER", "2"))
EAGER = object()                    # cache marker: this key could not be recorded, run the region eagerly
_STATS = {"recorded": 0, "replayed": 0, "poisoned": 0, "calls_replayed": 0}
_WARNED = set()


_PACE = [False]


def fast():
    """The ``_ospfast`` module (None when it is not built: tapes are then unavailable and every region runs eagerly)."""
    f = _lib.lib()._fast
    if f is not None and not _PACE[0]:
        _PACE[0] = True
        ns = int(os.environ.get("OSP_TAPE_PACE_NS", "0"))
        if ns and hasattr(f, "tape_set_pace"):
            f.tape_set_pace(ns)
    return f


def available():
    return ENABLED and fast() is not None and hasattr(fast(), "tape_begin")


def recording():
    f = fast()
    return f is not None and hasattr(f, "tape_recording") and f.tape_recording()


def stats():
    return dict(_STATS)


_REGION_PACKS = [None]


def region_packs():
    """While a Recorder is active: the set of (parameter, pack) pairs the region being recorded has refreshed so far (kernels.py
    treats the first use of an epoch-cached weight pack inside a recorded region as a miss, so that every tape carries the refresh
    of every pack it reads); None outside a recording."""
    return _REGION_PACKS[0]


# ---------------------------------------------------------------------------------------------------- ATen interception
#: operators that launch nothing: allocation, views, metadata
_NO_KERNEL = {
    "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::new_empty", "aten::new_empty_strided", "aten::view",
    "aten::_unsafe_view", "aten::reshape", "aten::_reshape_alias", "aten::as_strided", "aten::slice", "aten::select", "aten::narrow",
    "aten::transpose", "aten::permute", "aten::t", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::detach", "aten::alias",
    "aten::unbind", "aten::split", "aten::split_with_sizes", "aten::chunk", "aten::view_as", "aten::record_stream", "aten::lift_fresh",
    "aten::is_same_size", "aten::sym_size", "aten::sym_stride", "aten::sym_numel", "aten::sym_storage_offset", "aten::size",
    "aten::stride", "aten::numel", "aten::dim", "aten::is_contiguous", "aten::is_pinned", "aten::expand_as", "aten::unflatten",
    "aten::flatten", "aten::movedim", "aten::swapaxes", "aten::_unsafe_index", "aten::is_nonzero_placeholder",
}


def _plain(t):
    return isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()


def _r_zeros(func, args, kwargs):
    """zeros / zeros_like / new_zeros / zero_ / fill_(0): hipMemsetAsync through the C ABI."""
    name = func._schema.name
    if name == "aten::zero_" or (name == "aten::fill_" and not isinstance(args[1], torch.Tensor) and float(args[1]) == 0.0):
        t = args[0]
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
            return NotImplemented
        _lib.call("osp_memset", t, 0, t.numel() * t.element_size())
        return t
    kw = dict(kwargs or {})
    if name == "aten::zeros":
        out = torch.empty(args[0], **kw)
    elif name == "aten::zeros_like":
        kw.pop("memory_format", None)
        out = torch.empty_like(args[0], **kw, memory_format=torch.contiguous_format)
    elif name == "aten::new_zeros":
        out = args[0].new_empty(args[1], **kw)
    else:
        return NotImplemented
    if not out.is_cuda:
        return func(*args, **(kwargs or {}))
    if out.numel():
        _lib.call("osp_memset", out, 0, out.numel() * out.element_size())
    return out


def _bcast_mode(x, y):
    """osp_ew_mul mode of y against x (contiguous f32): 0 same shape, 1 one value per row, 2 one value per column; None otherwise."""
    if tuple(x.shape) == tuple(y.shape):
        return 0, 1
    if y.dim() <= x.dim() and x.dim() >= 1:
        ys = (1,) * (x.dim() - y.dim()) + tuple(y.shape)
        inner = x.shape[-1]
        if ys[-1] == 1 and tuple(ys[:-1]) == tuple(x.shape[:-1]):
            return 1, inner
        if ys[-1] == inner and all(v == 1 for v in ys[:-1]):
            return 2, inner
    return None, None


def _r_addsub(func, args, kwargs):
    name = func._schema.name
    a, b = args[0], args[1]
    alpha = float((kwargs or {}).get("alpha", 1.0))
    if name.startswith("aten::sub"):
        alpha = -alpha
    inplace = name.endswith("_")
    if isinstance(b, torch.Tensor) and _plain(a) and _plain(b) and tuple(a.shape) == tuple(b.shape):
        out = a if inplace else torch.empty_like(a)
        _lib.call("osp_ew_axpby", a, b, out, a.numel(), 1.0, alpha)
        return out
    if _plain(a) and not isinstance(b, torch.Tensor):
        out = a if inplace else torch.empty_like(a)
        _lib.call("osp_ew_axpby", a, None, out, a.numel(), 1.0, alpha * float(b))
        return out
    return NotImplemented


def _r_mul(func, args, kwargs):
    name = func._schema.name
    a, b = args[0], args[1]
    inplace = name.endswith("_")
    if isinstance(a, torch.Tensor) and not isinstance(b, torch.Tensor) and _plain(a):
        out = a if inplace else torch.empty_like(a)
        _lib.call("osp_ew_axpby", a, None, out, a.numel(), float(b), 0.0)
        return out
    if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)):
        return NotImplemented
    if a.numel() < b.numel() and not inplace:
        a, b = b, a
    if _plain(a) and b.is_cuda and b.dtype == torch.float32 and b.numel() == 1:           # times a device scalar
        if tuple(torch.broadcast_shapes(a.shape, b.shape)) != tuple(a.shape):            # (1,) * (1, 1): the result is not a's shape
            return NotImplemented
        out = a if inplace else torch.empty_like(a)
        _lib.call("osp_ew_scale_dev", a, b, out, a.numel(), 1.0)
        return out
    if _plain(a) and _plain(b):
        mode, inner = _bcast_mode(a, b)
        if mode is not None:
            out = a if inplace else torch.empty_like(a)
            _lib.call("osp_ew_mul", a, b, out, a.numel(), inner, mode)
            return out
    return NotImplemented


def _r_neg(func, args, kwargs):
    a = args[0]
    if _plain(a):
        out = torch.empty_like(a)
        _lib.call("osp_ew_axpby", a, None, out, a.numel(), -1.0, 0.0)
        return out
    return NotImplemented


def _r_div(func, args, kwargs):
    a, b = args[0], args[1]
    if (kwargs or {}).get("rounding_mode") is not None:           # floor / trunc division is not a scale: poison the recording
        return NotImplemented
    if _plain(a) and not isinstance(b, torch.Tensor):
        out = torch.empty_like(a)
        _lib.call("osp_ew_axpby", a, None, out, a.numel(), 1.0 / float(b), 0.0)
        return out
    return NotImplemented


def _r_clone(func, args, kwargs):
    a = args[0]
    if isinstance(a, torch.Tensor) and a.is_cuda and a.is_contiguous():
        out = torch.empty_like(a)
        if a.numel():
            _lib.call("osp_copy", out, a, a.numel() * a.element_size())
        return out
    return NotImplemented


def _r_copy_(func, args, kwargs):
    dst, src = args[0], args[1]
    if (isinstance(src, torch.Tensor) and dst.is_cuda and src.is_cuda and dst.dtype == src.dtype and dst.is_contiguous()
            and src.is_contiguous() and tuple(dst.shape) == tuple(src.shape)):
        if dst.numel():
            _lib.call("osp_copy", dst, src, dst.numel() * dst.element_size())
        return dst
    return NotImplemented


def _r_cat(func, args, kwargs):
    ts = list(args[0])
    dim = args[1] if len(args) > 1 else (kwargs or {}).get("dim", 0)
    if not ts or dim != 0 or not all(isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == ts[0].dtype
                                       and tuple(t.shape[1:]) == tuple(ts[0].shape[1:]) for t in ts):
        return NotImplemented
    out = torch.empty((sum(t.shape[0] for t in ts),) + tuple(ts[0].shape[1:]), device=ts[0].device, dtype=ts[0].dtype)
    row = 0
    for t in ts:
        if t.numel():
            _lib.call("osp_copy", out[row:], t, t.numel() * t.element_size())
        row += t.shape[0]
    return out


def _r_sum_mean(func, args, kwargs):
    """full reductions of small f32 tensors to a scalar (the per-utterance loss means)."""
    a = args[0]
    if len(args) > 1 or (kwargs and any(v is not None for v in kwargs.values())):
        return NotImplemented
    if _plain(a) and 0 < a.numel() <= (1 << 20):
        out = torch.empty((), device=a.device, dtype=torch.float32)
        scale = 1.0 / a.numel() if func._schema.name == "aten::mean" else 1.0
        _lib.call("osp_sum_scaled", a, a.numel(), scale, out)
        return out
    return NotImplemented


_REROUTE = {
    "aten::zeros": _r_zeros, "aten::zeros_like": _r_zeros, "aten::new_zeros": _r_zeros, "aten::zero_": _r_zeros, "aten::fill_": _r_zeros,
    "aten::add": _r_addsub, "aten::add_": _r_addsub, "aten::sub": _r_addsub, "aten::sub_": _r_addsub,
    "aten::mul": _r_mul, "aten::mul_": _r_mul, "aten::neg": _r_neg, "aten::div": _r_div,
    "aten::clone": _r_clone, "aten::copy_": _r_copy_, "aten::cat": _r_cat, "aten::sum": _r_sum_mean, "aten::mean": _r_sum_mean,
}


def _touches_gpu(args, kwargs):
    for a in list(args) + list((kwargs or {}).values()):
        if isinstance(a, torch.Tensor):
            if a.is_cuda:
                return True
        elif isinstance(a, (list, tuple)):
            if any(isinstance(t, torch.Tensor) and t.is_cuda for t in a):
                return True
    return False


class _AtenGuard(TorchDispatchMode):
    """Active while a region is recorded (see the module docstring)."""

    def __init__(self, rec):
        super().__init__()
        self.rec = rec

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if name in _NO_KERNEL:
            return func(*args, **(kwargs or {}))
        dev = (kwargs or {}).get("device")
        on_gpu = _touches_gpu(args, kwargs) or (dev is not None and torch.device(dev).type == "cuda")
        if not on_gpu:
            return func(*args, **(kwargs or {}))                 # host-side arithmetic launches nothing
        impl = _REROUTE.get(name)
        if impl is not None and (not self.rec.poisoned or os.environ.get("OSP_TAPE_DUMP", "0") == "1"):
            self.rec.rerouted += 1
            out = impl(func, args, kwargs)
            if out is not NotImplemented:
                return out
            self.rec.rerouted -= 1
        if os.environ.get("OSP_TAPE_DUMP", "0") == "1":            # every offending operator, not only the first (what to re-route next)
            shapes = [tuple(a.shape) if isinstance(a, torch.Tensor) else a for a in args][:4]
            print(f"[tape] {self.rec.label}: {func} {shapes} is not re-routed", flush=True)
        self.rec.poison(f"ATen operator {func} launched a kernel inside the region")
        return func(*args, **(kwargs or {}))


# ---------------------------------------------------------------------------------------------------- recording / replay
def _extent(t):
    return t.numel() * t.element_size()


class Region:
    """A recorded region: replay(inputs) -> the region's outputs (fresh aliases of persistent buffers)."""
    __slots__ = ("cap", "outs", "n_inputs", "ncalls", "meta", "extra")

    def __init__(self, cap, outs, n_inputs, ncalls, meta):
        self.cap, self.outs, self.n_inputs, self.ncalls, self.meta = cap, outs, n_inputs, ncalls, meta
        self.extra = None

    def replay(self, inputs):
        # what Recorder.__enter__ checked at record time holds for every replay: same byte extent and type, contiguous, on the device
        # (a same-shape view with other strides or another dtype would make the patched launches read the wrong memory silently)
        sig = self.meta["sig"]
        if len(inputs) != len(sig):
            raise ValueError(f"tape {self.meta['label']!r}: {len(inputs)} inputs given, {len(sig)} recorded")
        for t, want in zip(inputs, sig):
            if (t is None) != (want is None) or (t is not None and ((_extent(t), t.dtype) != want or not t.is_contiguous() or not t.is_cuda)):
                raise ValueError(f"tape {self.meta['label']!r}: an input differs from the recorded one (extent / dtype / layout)")
        bases = [0 if t is None else t.data_ptr() for t in inputs]
        rc = fast().tape_replay(self.cap, bases, _lib._STREAM_OVERRIDE[0] or _lib._raw_stream(_lib._cur_device()))
        if rc != 0:
            raise _lib.OspError(f"tape replay: {rc[1]} failed ({rc[0]}): {_lib.lib().cdll.osp_last_error().decode()}")
        _STATS["replayed"] += 1
        _STATS["calls_replayed"] += self.ncalls
        return _alias(self.outs)


def _alias(outs):
    if outs is None:
        return None
    if isinstance(outs, torch.Tensor):
        return outs.detach()
    return tuple(_alias(o) for o in outs)


class Recorder:
    """``with Recorder(inputs) as rec: outs = fn(*inputs)`` then ``rec.finish(outs)`` -> Region, or None when the recording was
    poisoned (the execution itself was complete and valid either way: a recording run IS a normal run)."""

    def __init__(self, inputs, label="", packs=None):
        self.inputs = list(inputs)
        self.label = label
        #: weight packs this region may take as fresh without a refresh of its own: those refreshed by the region whose replay
        #: always precedes this one's in the same optimizer epoch (a Segment's backward inherits its forward's)
        self.packs = set(packs) if packs is not None else set()
        self.poisoned = None
        self.rerouted = 0
        self.cap = None
        self._guard = None

    def poison(self, why):
        if self.poisoned is None:
            self.poisoned = why

    def __enter__(self):
        f = fast()
        for t in self.inputs:
            if t is not None and not (t.is_cuda and t.is_contiguous()):
                self.poison("a region input is not a contiguous device tensor")
        bases = [0 if t is None else t.data_ptr() for t in self.inputs]
        sizes = [0 if t is None else _extent(t) for t in self.inputs]
        f.tape_begin(bases, sizes, _lib._STREAM_OVERRIDE[0] or _lib._raw_stream(_lib._cur_device()))
        for t in self.inputs:                                    # inputs stay alive only through the caller; nothing to keep
            pass
        _REGION_PACKS[0] = self.packs
        self._guard = _AtenGuard(self)
        self._guard.__enter__()
        return self

    def __exit__(self, et, ev, tb):
        f = fast()
        _REGION_PACKS[0] = None
        self._guard.__exit__(et, ev, tb)
        if et is not None:
            f.tape_abort()
            return False
        self.cap = f.tape_end()
        return False

    def finish(self, outs):
        if self.cap is None:
            return None
        if self.poisoned is not None:
            _STATS["poisoned"] += 1
            key = (self.label, self.poisoned)
            if key not in _WARNED:
                _WARNED.add(key)
                warnings.warn(f"optispeech_amd.tape: region {self.label!r} stays eager: {self.poisoned}")
            self.cap = None
            return None
        ncalls, npatches, names, streams = fast().tape_info(self.cap)
        _STATS["recorded"] += 1
        if os.environ.get("OSP_TAPE_DUMP", "0") == "1":
            import collections
            per = collections.Counter("current" if st is None else hex(st) for st in streams)
            print(f"[tape] {self.label}: {ncalls} calls, {npatches} patches, {self.rerouted} re-routed ATen ops, streams {dict(per)}", flush=True)
            if os.environ.get("OSP_TAPE_DUMP_CALLS", "0") == "1":
                for n, st in zip(names, streams):
                    print(f"[tape]    {'current' if st is None else hex(st):>14s} {n}", flush=True)
        # aliases taken NOW (inside the caller's Function.forward the outputs carry no autograd history yet): the objects handed to
        # the caller get a grad_fn later, and holding those would pin the first step's graph
        return Region(self.cap, _alias(outs), len(self.inputs), ncalls,
                      {"patches": npatches, "rerouted": self.rerouted, "label": self.label, "names": names,
                       "sig": [None if t is None else (_extent(t), t.dtype) for t in self.inputs]})


def run(cache, key, inputs, fn, label=""):
    """Run region ``fn(*inputs)`` through the tape cache ``cache`` (a dict owned by the caller): record on the first call with
    ``key``, replay afterwards.  ``fn`` returns a tensor, a (nested) tuple of tensors / None, or None."""
    if not available() or recording() or _lib._RECORD[0] is not None or torch.cuda.is_current_stream_capturing():
        return fn(*inputs)                                       # (inside a hipGraph capture the graph owns the memory: stay eager)
    ent = cache.get(key)
    if ent is None:
        if len(cache) >= 32:                                     # each region owns its activation buffers: bound the set
            cache.pop(next(iter(cache)))
        with Recorder(inputs, label) as rec:
            outs = fn(*inputs)
        region = rec.finish(outs)
        cache[key] = region if region is not None else EAGER
        return outs
    if ent is EAGER:
        return fn(*inputs)
    return ent.replay(inputs)


# ---------------------------------------------------------------------------------------------------- taped autograd segments
class Segment:
    """A differentiable sub-graph of the step -- the acoustic model, the vocoder -- as ONE autograd node whose forward and backward
    are call tapes.

    ``fn(*inputs) -> tuple of tensors`` builds an ordinary autograd graph out of this package's Functions (any number of nodes,
    side streams through the C ABI, parameter gradients accumulated straight into the gradient arena).  The first call with a given
    ``key`` RUNS it that way -- forward under a recorder, then, when the step's backward arrives, the inner graph's backward under a
    second recorder -- so the recording step computes exactly what an eager step computes.  Later calls replay the two tapes: no
    inner graph, no Python per kernel.  The inputs receive no gradient (the acoustic model's are data, the vocoder's is the
    detached decoder segment).  A recording that met something it cannot hold leaves the key eager: ``fn`` then simply runs under
    the caller's autograd, every step."""

    def __init__(self, fn, label):
        self.fn, self.label = fn, label
        self.cache = {}
        self.seen = {}
        self.anchor = None

    def __call__(self, key, *inputs):
        if (not available() or recording() or not torch.is_grad_enabled() or not inputs[0].is_cuda
                or torch.cuda.is_current_stream_capturing() or self.cache.get(key) is EAGER):
            return self.fn(*inputs)
        if key not in self.cache and SEGMENT_RECORD_AFTER > 1:
            # a segment's key is built from the batch's SHAPES and a recorded segment owns a full set of activations: record a key only
            # once it has come back (ragged real batches would otherwise re-record almost every step, never replay, and pin
            # up to 8 activation sets); `seen` holds keys only, bounded
            n = self.seen.pop(key, 0) + 1
            if n < SEGMENT_RECORD_AFTER:
                if len(self.seen) >= 256:
                    self.seen.pop(next(iter(self.seen)))
                self.seen[key] = n
                return self.fn(*inputs)
        if self.anchor is None or self.anchor.device != inputs[0].device:
            self.anchor = torch.zeros(1, device=inputs[0].device, requires_grad=True)     # gives the node's outputs a grad_fn
        return _SegmentFn.apply(self, key, self.anchor, *inputs)


class _SegmentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, key, anchor, *inputs):
        ctx.seg, ctx.key = seg, key
        ctx.set_materialize_grads(False)
        inputs = [t.contiguous() if isinstance(t, torch.Tensor) else t for t in inputs]
        # the backward reads the segment's INPUTS too (token ids, lengths: embedding / loss / mask kernels): they are declared inputs
        # of the backward tape as well, or it would keep reading the tensors of the recording step (round 5: a replay on another
        # ragged batch of the same shape gave 2.5e-2 wrong acoustic-model gradients; invisible while every step saw the same tensors)
        ctx.fwd_inputs = inputs
        ent = seg.cache.get(key)
        if ent is not None:
            outs = ent[0].replay(inputs)
            ctx.inner = None
            ctx.diff = ent[2]
            ctx.ent = ent                                          # (the cache may evict the key before the backward arrives)
        else:
            if len(seg.cache) >= 8:
                seg.cache.pop(next(iter(seg.cache)))
            with Recorder(inputs, seg.label + " forward") as rec:
                with torch.enable_grad():
                    inner = tuple(seg.fn(*[t.detach() if isinstance(t, torch.Tensor) else t for t in inputs]))
            diff = tuple(i for i, o in enumerate(inner) if isinstance(o, torch.Tensor) and o.requires_grad)
            region = rec.finish(tuple(o.detach() if isinstance(o, torch.Tensor) else o for o in inner))
            ctx.inner = (inner, region)
            ctx.fwd_packs = rec.packs
            ctx.diff = diff
            outs = tuple(o.detach() if isinstance(o, torch.Tensor) else o for o in inner)
        nd = [o for i, o in enumerate(outs) if isinstance(o, torch.Tensor) and i not in ctx.diff]
        if nd:
            ctx.mark_non_differentiable(*nd)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        seg, key = ctx.seg, ctx.key
        gin = [None if grads[i] is None else grads[i].contiguous() for i in ctx.diff]
        pattern = tuple(None if g is None else (tuple(g.shape), g.dtype) for g in gin)
        nin = len(ctx.needs_input_grad)
        if ctx.inner is None:
            fwd, bwd, diff, pat = ctx.ent
            if pat != pattern:
                raise RuntimeError(f"taped segment {seg.label!r}: the step asks for gradients of a different set of outputs than the "
                                   f"recorded backward has ({pattern} vs {pat}); give such a step its own key")
            bwd.replay(gin + list(ctx.fwd_inputs))
            ctx.fwd_inputs = None
            return (None,) * nin
        inner, fwd = ctx.inner
        ctx.inner = None
        outs = [inner[i] for i in ctx.diff]
        pairs = [(o, g) for o, g in zip(outs, gin) if g is not None]
        if fwd is None:                                            # the forward recording was poisoned: plain nested backward
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
            seg.cache[key] = EAGER
            return (None,) * nin
        from . import ops as _ops
        outer = _ops.nested_backward_begin()                       # the segment joins its own weight-gradient side streams: on the tape
        try:
            with Recorder(gin + list(ctx.fwd_inputs), seg.label + " backward", packs=ctx.fwd_packs) as rec:
                torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        finally:
            _ops.nested_backward_end(outer)
        bwd = rec.finish(None)
        seg.cache[key] = (fwd, bwd, ctx.diff, pattern) if bwd is not None else EAGER
        return (None,) * nin
