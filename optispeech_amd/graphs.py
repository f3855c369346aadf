"""hipGraph capture / replay of the training step and of the batched decode (BASELINE.json configs[1] / configs[4]).

Why: the eager step enqueues ~1 400 launches from Python (autograd nodes, C-ABI calls, torch glue) and needs 21-23 ms of host
time for ~20 ms of GPU work -- the step was host-bound.  A captured step replays the same launches,
streams and dependencies with one ``hipGraphLaunch`` per segment.

What makes the step capturable:
  * zero host synchronisations inside it (lengths, MAS, segment starts, clip factor all stay on the device);
  * everything that changes from step to step lives in DEVICE memory and is refreshed by one small host->device copy in front
    of each replay: the Philox seed of the fused dropout sites (``seed_dev`` of osp_layernorm_* / osp_text_embed_* /
    osp_attn_softmax_*), the AdamW step counts and learning rates (``step_dev`` / ``lr_dev`` of osp_adamw_clip).  The host keeps
    computing the cosine schedule in double precision exactly as the eager path (and the reference) does;
  * inputs are copied into static buffers; outputs (the logged scalars) are static device tensors;
  * torch's own generator (drop-path Bernoulli draws, segment starts) is graph-safe (Philox offset advanced per replay).

Segments: on one GPU the whole step is ONE graph (the eight sub-discriminator streams, the vocoder stream and the CTC side stream
fork from and re-join the capture stream exactly as in eager mode).  Under data parallelism the gradient all-reduces stay
OUTSIDE the graphs (RCCL calls between graph launches), so the step is cut at the collectives and the stages are re-ordered
so that each all-reduce overlaps a graph that does not need its result:

    [G forward] -> [D forward+backward] -> all-reduce(D grads) || [G backward] -> all-reduce(G grads) || [AdamW(D)] -> [AdamW(G)]

The discriminator phase reads the discriminator weights of the step's start and ``wav_hat.detach()`` only, the generator's
backward reads the same (frozen) discriminator weights, and the two optimisers own disjoint parameters: every stage sees exactly
the values it sees in the reference's order (base_lightning_module.py:78-126), only the issue order differs.
"""
import contextlib
import gc

import torch

from . import rng, values


@contextlib.contextmanager
def no_gc_during_capture():
    """Keep Python's CYCLIC garbage collector out of a hipGraph capture.

    A collection that happens to trigger inside a capture (any allocation can start one) finalises whatever cyclic garbage is
    around -- e.g. an earlier StepGraphs / model pair with its pinned scalar ring, events and captured graphs -- and those
    finalisers make HIP calls that are illegal while a stream is capturing (event queries of the pinned-memory allocator, graph
    / event destruction).  The error surfaces inside a C++ destructor, i.e. as std::terminate: the interpreter dies with SIGABRT.
    That is what killed round 2's driver run of the GPU suite (reproduced in round 3: the faulthandler stack shows
    'Garbage-collecting' under a kernel launch inside ``with torch.cuda.graph(...)``, profiles/r03_gputest_abort_stack.txt).
    Reference-counted frees are deterministic and stay as they are; only the cycle collector is held back until the capture ended."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _shape_key(model, batch, train_d):
    sig = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if torch.is_tensor(v) or hasattr(v, "shape"))
    return (sig, bool(train_d), bool(model.training), bool(model.replay_disc_forward))


class StepGraphs:
    """One captured training step for one (batch signature, regime)."""

    def __init__(self, model, batch, warmup=2):
        self.model = model
        self.dev = model.device
        opt_g, opt_d = model.optimizers()
        self.red_g, self.red_d = model._reducers
        self.segmented = self.red_g.active or bool(getattr(model, "graph_force_segments", False))    # test hook: the DP stage order on one GPU
        # static inputs
        self.static = {}
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k] = v.detach().to(self.dev).clone()
            elif hasattr(v, "shape") and hasattr(v, "dtype"):                 # host numpy ``wav`` of the reference's collate
                self.static[k] = torch.from_numpy(v).to(self.dev, dtype=torch.float32)
            else:
                self.static[k] = v
        # per-step scalars in device memory (+ pinned host mirrors): [seed, step_g, step_d] and [lr_g, lr_d]
        self.si = torch.zeros(3, dtype=torch.int64, device=self.dev)
        self.sf = torch.zeros(2, dtype=torch.float32, device=self.dev)
        # the host runs ahead of the device by several replays: a ring of pinned slots, each re-used only after its copy ran
        self._ring = [(torch.zeros(3, dtype=torch.int64).pin_memory(), torch.zeros(2, dtype=torch.float32).pin_memory(), [None])
                      for _ in range(16)]
        self._slot = 0
        self.graphs, self.between = [], []
        self._capture(batch, warmup)

    # ------------------------------------------------------------------------------------------ scalars
    def _push_scalars(self):
        """Values the NEXT step uses, exactly what the eager path would pass as launch arguments."""
        m = self.model
        opt_g, opt_d = m.optimizers()
        hi, hf, ev = self._ring[self._slot]
        self._slot = (self._slot + 1) % len(self._ring)
        if ev[0] is not None:
            ev[0].synchronize()
        hi[0], hi[1], hi[2] = rng.host_seed() + 1, opt_g.step_count + 1, opt_d.step_count + 1
        hf[0], hf[1] = float(opt_g.lr), float(opt_d.lr)
        self.si.copy_(hi, non_blocking=True)
        self.sf.copy_(hf, non_blocking=True)
        if not torch.cuda.is_current_stream_capturing():
            ev[0] = torch.cuda.current_stream().record_event()

    def _advance_host(self, train_d):
        """The host-side bookkeeping of one step (what the stage functions do in eager mode)."""
        m = self.model
        opt_g, opt_d = m.optimizers()
        sg, sd = m.lr_schedulers()
        rng.advance()
        opt_g.step_count += 1
        sg.step()
        m.global_step += 1
        if train_d:
            opt_d.step_count += 1
            sd.step()
            m.global_step += 1
        from . import values
        values.bump_param_epoch()

    def _host_state(self):
        m = self.model
        opt_g, opt_d = m.optimizers()
        sg, sd = m.lr_schedulers()
        from . import values
        return (dict(rng._state), opt_g.step_count, opt_d.step_count, opt_g.lr, opt_d.lr, sg.last_step, sd.last_step, m.global_step,
                values.param_epoch())

    def _restore_host_state(self, s):
        m = self.model
        opt_g, opt_d = m.optimizers()
        sg, sd = m.lr_schedulers()
        rng._state.update(s[0])
        opt_g.step_count, opt_d.step_count, opt_g.lr, opt_d.lr, sg.last_step, sd.last_step, m.global_step = s[1:8]

    # ------------------------------------------------------------------------------------------ stages
    def _segments(self, st):
        """[(stage callables of one graph, collective hook run right after the graph is launched)]."""
        m, b = self.model, self.static
        opt_g, opt_d = m.optimizers()
        if not st.train_d:
            # pre-training regime: nothing to overlap the G all-reduce with -- start it and make the stream the AdamW(G)
            # graph is launched on wait for it (without the wait the update read partially reduced gradients)
            return [([lambda: m._stage_g_forward(st, b), lambda: m._stage_g_backward(st)],
                     lambda: (self.red_g.start(opt_g.arena.grad), self.red_g.wait())),
                    ([lambda: m._stage_opt_g(st)], None)] if self.segmented else \
                   [([lambda: m._stage_g_forward(st, b), lambda: m._stage_g_backward(st), lambda: m._stage_opt_g(st)], None)]
        if not self.segmented:
            return [([lambda: m._stage_g_forward(st, b), lambda: m._stage_g_backward(st), lambda: m._stage_d(st, b),
                      lambda: m._stage_opt_g(st), lambda: m._stage_opt_d(st)], None)]
        return [([lambda: m._stage_g_forward(st, b)], None),
                ([lambda: self._fresh_ready_event(st), lambda: m._stage_d(st, b)], lambda: self.red_d.start(opt_d.arena.grad)),
                ([lambda: m._stage_g_backward(st)], lambda: (self.red_d.wait(), self.red_g.start(opt_g.arena.grad))),
                ([lambda: m._stage_opt_d(st)], lambda: self.red_g.wait()),
                ([lambda: m._stage_opt_g(st)], None)]

    @staticmethod
    def _fresh_ready_event(st):
        """The 'inputs ready' event prepare_disc_inputs recorded belongs to the previous segment's capture: a stream of this
        capture cannot wait on it.  Dropping it makes the discriminator phase record its own (its inputs are complete anyway:
        the previous graph has been launched in full)."""
        if st.pre is not None:
            st.pre = (st.pre[0], None)

    def _run_eager(self):
        """One step through the same stage order, eagerly (warm-up before the capture: allocator pools, lazily created streams,
        caches, autotuned launch attributes)."""
        m = self.model
        st = m._new_step_state()
        self._push_scalars()
        for stages, hook in self._segments(st):
            for f in stages:
                f()
            if hook is not None:
                hook()
        m.last_logs = st.logs
        # the stage functions advanced step counts / schedules / global_step themselves; the seed is ours
        rng.advance()
        return st.train_d

    def _capture(self, batch, warmup):
        m = self.model
        opt_g, opt_d = m.optimizers()
        keep_pipe, m.pipeline_steps = m.pipeline_steps, False
        m.join()
        rng.use_device_seed(self.si[0:1])
        keep_ranges = (getattr(self.red_g, "eager_ranges", True), getattr(self.red_d, "eager_ranges", True))
        self.red_g.eager_ranges = self.red_d.eager_ranges = False      # collectives stay between the graphs, never inside a capture
        opt_g.dev_scalars = (self.sf[0:1], self.si[1:2])
        opt_d.dev_scalars = (self.sf[1:2], self.si[2:3])
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(cur)
        try:
            with torch.cuda.stream(s):
                # warm-up through the same code path (allocator pools, lazily created streams, caches), then roll the model
                # back: weights, AdamW moments, counters and torch's generator -- a capture advances nothing
                saved = self._host_state()
                snap = [t.clone() for o in (opt_g, opt_d) for t in (o.arena.data, o.exp_avg, o.exp_avg_sq)]
                gen_state = torch.cuda.get_rng_state(self.dev)
                for _ in range(max(1, warmup)):
                    self._run_eager()
                torch.cuda.synchronize(self.dev)
                for dst, src in zip([t for o in (opt_g, opt_d) for t in (o.arena.data, o.exp_avg, o.exp_avg_sq)], snap):
                    dst.copy_(src)
                del snap
                torch.cuda.set_rng_state(gen_state, self.dev)
                self._restore_host_state(saved)
                from . import values
                values.bump_param_epoch()                      # weight packs cached during the warm-up are stale again
                torch.cuda.synchronize(self.dev)
                st = m._new_step_state()
                self.train_d = st.train_d
                self._push_scalars()
                pool = None
                for stages, hook in self._segments(st):
                    g = torch.cuda.CUDAGraph()
                    with no_gc_during_capture(), torch.cuda.graph(g, pool=pool, stream=s, capture_error_mode="thread_local"):
                        for f in stages:
                            f()
                    pool = g.pool() if pool is None else pool
                    self.graphs.append(g)
                    self.between.append(hook)
                self.logs = st.logs
                self._restore_host_state(saved)
            cur.wait_stream(s)
        finally:
            self.red_g.eager_ranges, self.red_d.eager_ranges = keep_ranges
            rng.use_device_seed(None)
            opt_g.dev_scalars = opt_d.dev_scalars = None
            m.pipeline_steps = keep_pipe
        for p in m._disc_params():
            p.requires_grad_(True)

    # ------------------------------------------------------------------------------------------ replay
    def replay(self, batch):
        m = self.model
        for k, v in batch.items():
            dst = self.static.get(k)
            if torch.is_tensor(dst) and v is not dst:
                src = v if torch.is_tensor(v) else torch.from_numpy(v)
                dst.copy_(src, non_blocking=True)
        self._push_scalars()
        for g, hook in zip(self.graphs, self.between):
            g.replay()
            if hook is not None:
                hook()
        self._advance_host(self.train_d)
        m.last_logs = self.logs


def graphed_training_step(model, batch, batch_idx=0):
    """``OptiSpeech.training_step`` through captured graphs: captures on the first call with a given batch signature / regime
    (the capture's warm-up steps are rolled back: one call is one step), replays on every call."""
    train_d = model.global_step >= model.train_args.pretraining_steps
    key = _shape_key(model, batch, train_d)
    sg = model._step_graphs.get(key)
    if sg is None:
        if len(model._step_graphs) >= 4:                                   # each capture owns its activation pool
            model._step_graphs.pop(next(iter(model._step_graphs)))
        sg = model._step_graphs[key] = StepGraphs(model, batch, warmup=int(getattr(model, "graph_warmup_steps", 2)))
    sg.replay(batch)


# =================================================================================================== graphed segments
# The whole-step graph above loses to the eager step on ROCm 7.2 because the runtime executes the captured multi-stream graph
# with little concurrency.  What the eager step pays instead is host time: ~22 ms of Python / autograd / dispatch per step, about
# two thirds of it for the acoustic model and the vocoder -- long, essentially LINEAR launch chains.  Graphed segments take
# exactly those chains out of Python and leave the multi-stream part (the eight discriminator stacks) eager:
#
#   forward graph + backward graph per segment, wrapped in an autograd Function (what torch.cuda.make_graphed_callables does,
#   adapted to this package's direct-to-arena parameter gradients): the forward replays graph F from static inputs into static
#   outputs, the backward copies the incoming output gradients into static buffers and replays graph B, whose captured kernels
#   accumulate the parameter gradients straight into the flat gradient arena (fixed addresses).
class GraphedSegment:
    def __init__(self, fn, inputs, warmup=2):
        dev = inputs[0].device
        self.static_in = [t.detach().clone() for t in inputs]
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)          # gives the Function's outputs a grad_fn
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(warmup):
                outs = fn(*self.static_in)
                go = [o for o in outs if o.requires_grad]
                # zero output gradients: the warm-up backward adds zeros to the gradient arena
                torch.autograd.backward(go, [torch.zeros_like(o) for o in go])
            del outs, go
            torch.cuda.synchronize(dev)
            # derived weight packs cached by the warm-up (kernels.param_bf16 / _param_pack, weight-norm packs) must be re-made INSIDE
            # the capture: the graph then refreshes them from the live weights on every replay, in its own pool -- a pack made
            # eagerly would be read by the replays long after the next optimizer epoch has freed it
            values.bump_param_epoch()
            self.gf = torch.cuda.CUDAGraph()
            with no_gc_during_capture(), torch.cuda.graph(self.gf, stream=s, capture_error_mode="thread_local"):
                outs = fn(*self.static_in)
            self.outs = tuple(outs)
            self.grad_idx = [i for i, o in enumerate(self.outs) if o.requires_grad]
            self.static_go = [torch.zeros_like(self.outs[i]) for i in self.grad_idx]
            self.gb = torch.cuda.CUDAGraph()
            with no_gc_during_capture(), torch.cuda.graph(self.gb, pool=self.gf.pool(), stream=s, capture_error_mode="thread_local"):
                torch.autograd.backward([self.outs[i] for i in self.grad_idx], self.static_go)
            self.outs = tuple(o.detach() for o in self.outs)
            values.bump_param_epoch()                          # and nothing eager keeps using a pack that lives in the graph's pool
        cur.wait_stream(s)

    def __call__(self, *inputs):
        return _GraphedFn.apply(self, self.anchor, *inputs)


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, anchor, *inputs):
        for st, t in zip(seg.static_in, inputs):
            if t is not st:
                st.copy_(t, non_blocking=True)
        seg.gf.replay()
        ctx.seg = seg
        ctx.set_materialize_grads(False)
        outs = tuple(o.view_as(o) for o in seg.outs)
        nd = [o for i, o in enumerate(outs) if i not in seg.grad_idx]
        if nd:
            ctx.mark_non_differentiable(*nd)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        seg = ctx.seg
        for k, i in enumerate(seg.grad_idx):
            if grads[i] is None:
                seg.static_go[k].zero_()
            else:
                seg.static_go[k].copy_(grads[i])
        seg.gb.replay()
        return (None, None) + (None,) * len(seg.static_in)


class GeneratorSegments:
    """Acoustic model and vocoder of one OptiSpeech model as two graphed segments for one batch signature."""

    def __init__(self, model, tensors):
        gen = model.generator
        dev = model.device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._ring = [(torch.zeros(1, dtype=torch.int64).pin_memory(), [None]) for _ in range(16)]
        self._slot = 0
        self.push_seed()

        def am(x, x_lengths, mel, mel_lengths, pitches, energies):
            o = gen._forward_am(x, x_lengths, mel, mel_lengths, pitches, energies, None, None, vocoder_hook=None)
            return (o["loss"], o["align_loss"], o["duration_loss"], o["pitch_loss"], o["energy_loss"], o["_aux"]["segment"], o["start_idx"])

        rng.use_device_seed(self.seed_dev)
        gen_state = torch.cuda.get_rng_state(dev)               # the warm-up runs draw drop-path masks: a capture advances nothing
        try:
            self.am = GraphedSegment(am, tensors)
            self.voc = GraphedSegment(lambda seg: (gen.vocoder(seg, f0=None),), (self.am.outs[5],))
        finally:
            rng.use_device_seed(None)
            torch.cuda.set_rng_state(gen_state, dev)
        self.segment_size = int(self.am.outs[5].shape[1])

    def push_seed(self):
        """The seed the step about to run uses (rng.advance() has been called by training_step), into device memory."""
        h, ev = self._ring[self._slot]
        self._slot = (self._slot + 1) % len(self._ring)
        if ev[0] is not None:
            ev[0].synchronize()
        h[0] = rng.host_seed()
        self.seed_dev.copy_(h, non_blocking=True)
        ev[0] = torch.cuda.current_stream().record_event()
