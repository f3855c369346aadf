"""Host-side mirror of optispeech/model/optispeech.py + base_lightning_module.py: the ``OptiSpeech`` module with
``training_step`` (manual two-optimiser GAN step), ``configure_optimizers``, ``synthesise``/``synthesize``,
``prepare_input`` and ``load_from_checkpoint`` -- no Lightning / Hydra dependency on the GPU box.

Trainer integration: Lightning calls ``training_step(batch, batch_idx)`` with manual optimisation; here the
module owns its optimisers (``configure_optimizers()`` is called lazily), the step counter Lightning would keep
(``global_step`` = number of optimiser steps taken, both optimisers counted -- which is what
``self.global_step >= pretraining_steps`` compares against upstream) and the data-parallel gradient reducer.
"""
import contextlib
import io
import pickle
from functools import partial
from types import SimpleNamespace

import os

import numpy as np
import torch
from torch import nn

from .. import kernels as K
from .. import ops
from .. import rng
from .. import tape as _tape
from ..dp import GradReducer
from ..optim import CosineWarmupSchedule, FusedAdamW
from ..values import InferenceInputs, InferenceOutputs


class IdsTextProcessor:
    """Minimal stand-in for optispeech.text.TextProcessor (espeak/piper_phonemize are host string processing and out
    of scope): accepts pre-tokenised phoneme ids."""
    languages = ["en-us"]
    is_multi_language = False
    num_languages = 1

    def __call__(self, text, lang=None, split_sentences=True):
        if isinstance(text, str):
            sents = [[int(t) for t in s.split()] for s in text.split("|") if s.strip()]
        else:
            sents = [list(s) for s in (text if text and isinstance(text[0], (list, tuple, np.ndarray)) else [text])]
        if split_sentences:
            return sents, [" ".join(map(str, s)) for s in sents]
        flat = [t for s in sents for t in s]
        return flat, " ".join(map(str, flat))


def default_args(batch_size=32, feature_extractor=None):
    from ..config import FeatureExtractorArgs
    train_args = SimpleNamespace(cache_generator_outputs=True, gradient_clip_val=10, gradient_accumulate_batches=None,
                                 pretraining_steps=1000, evaluate_periodicity=False, evaluate_utmos=False,
                                 evaluate_pesq=False)
    data_args = SimpleNamespace(name="synthetic", num_speakers=1, text_processor=IdsTextProcessor(),
                                feature_extractor=feature_extractor or FeatureExtractorArgs(), batch_size=batch_size,
                                data_statistics=None)
    inference_args = SimpleNamespace(d_factor=1.1, p_factor=1.6, e_factor=1.2)
    return train_args, data_args, inference_args


_SYNC_AFTER_G = __import__("os").environ.get("OSP_SYNC_AFTER_G", "0") == "1"
#: backward() runs on the calling thread instead of the engine's per-device worker thread: one device per process, so the worker
#: only adds a thread hand-over and GIL traffic per pass (host enqueue 20.2 -> 17.9 ms per step at B = 32; OSP_AUTOGRAD_MT=1 restores it)
_AUTOGRAD_MT = __import__("os").environ.get("OSP_AUTOGRAD_MT", "0") == "1"
_D_AFTER_G = __import__("os").environ.get("OSP_D_AFTER_G", "0") == "1"
#: pipelined steps: the discriminator phase starts when the generator's backward has left the discriminator stacks (not at its end),
#: and the generator's optimiser update is issued BEFORE the discriminator phase (so that it -- and the next step's acoustic-model
#: forward behind it -- does not queue behind discriminator-phase packets of a shared hardware queue).  MEASURED AND LEFT OFF (round 5,
#: DESIGN.md section 9): 15.45 ms (early start alone: the phase still starts when the backward ends) / 16.4-16.7 ms (update first)
#: against 15.45 ms, with every placement of the phase's stream on the hardware queues tried (profiles/r05_early_d_ab.txt)
_EARLY_D = __import__("os").environ.get("OSP_EARLY_D", "0") == "1"
#: pipelined discriminator phase: per-stack forward -> loss term -> backward without a phase-wide join (round 5).  MEASURED AND LEFT OFF:
#: 15.09-15.11 vs 14.92-14.99 ms per step (eight small backward() calls cost the host more than the missing join gives the device;
#: DESIGN.md section 9); OSP_INLINE_D=1 turns it on
_INLINE_D = __import__("os").environ.get("OSP_INLINE_D", "0") == "1"
_G_OPT_FIRST = __import__("os").environ.get("OSP_G_OPT_FIRST", "0") == "1"


class OptiSpeech(nn.Module):
    def __init__(self, dim, generator, vocoder, discriminator, train_args, data_args, inference_args, optimizer=None,
                 scheduler=None):
        super().__init__()
        if (train_args.gradient_accumulate_batches is not None) and (train_args.gradient_accumulate_batches <= 0):
            raise ValueError("gradient_accumulate_batches should be a positive number")          # optispeech.py:29-30
        if data_args.num_speakers < 1:
            raise ValueError("num_speakers should be a positive integer >= 1")                  # optispeech.py:32-33
        self.train_args, self.data_args, self.inference_args = train_args, data_args, inference_args
        self.text_processor = data_args.text_processor
        self.num_speakers = data_args.num_speakers
        self.sample_rate = data_args.feature_extractor.sample_rate
        self.hop_length = data_args.feature_extractor.hop_length
        self.automatic_optimization = False
        self.hparams = SimpleNamespace(optimizer=optimizer, scheduler=scheduler)
        self.generator = generator(dim=dim, vocoder=vocoder, feature_extractor=data_args.feature_extractor,
                                   data_statistics=data_args.data_statistics, num_speakers=data_args.num_speakers,
                                   num_languages=self.text_processor.num_languages)
        self.discriminator = discriminator(feature_extractor=data_args.feature_extractor)
        self.global_step = 0
        self.max_steps = 2_000_000
        self._opts = None
        self.share_real_pass = os.environ.get("OSP_SHARE_REAL", "0") == "1"
        #: issue the discriminator phase from its own stream so that consecutive steps overlap (see training_step); off by
        #: default because code that reads discriminator parameters from another stream must then call join() first
        self.pipeline_steps = os.environ.get("OSP_PIPELINE_STEPS", "0") == "1"
        #: opt-in: the discriminator phase replays the backward on the forward the generator phase already ran (same waves,
        #: same weights inside one step -> identical activations) instead of evaluating the discriminators a second time.
        #: Off by default: the benchmarked step does every forward the reference does.
        self.replay_disc_forward = os.environ.get("OSP_DISC_REPLAY", "0") == "1"
        #: replay the step from hipGraphs (optispeech_amd/graphs.py): captured on the first step of each (batch shape,
        #: regime), replayed afterwards.  Off by default in the library (a captured step needs fixed shapes); bench.py
        #: turns it on
        self.graph_steps = os.environ.get("OSP_GRAPH_STEPS", "0") == "1"
        self._step_graphs = {}
        #: acoustic model + vocoder (forward and backward) replayed from captured hipGraphs inside the otherwise eager,
        #: multi-stream step (optispeech_amd/graphs.py: GeneratorSegments): removes ~2/3 of the step's host time
        self.graph_segments = os.environ.get("OSP_GRAPH_SEGMENTS", "0") == "1"
        self._gen_segments = {}
        #: acoustic model + vocoder as TAPED segments (optispeech_amd/tape.py: one autograd node each, forward and backward
        #: replayed from recorded C-ABI call lists on the eager multi-stream schedule): host enqueue 13.8 -> 7.5 ms per step.  Default
        #: since round 5: in round 4 the device needed 0.6-3 ms longer for the same kernels with the vocoder taped (never explained);
        #: with round 5's weight-gradient kernels and launch order the two schedules measure the same (15.2-15.6 ms, three same-box
        #: pairs, profiles/r05_tape_segments_ab.txt), and at the small per-rank batches of strong scaling the taped host is what the
        #: step time is (B = 8: 8.8 vs 14.1 ms).  OSP_TAPE_SEGMENTS=0 restores the eager generator.  The sub-discriminator stacks
        #: are taped regardless of this switch.
        self.tape_segments = os.environ.get("OSP_TAPE_SEGMENTS", "1") != "0"
        self._tape_am = self._tape_voc = None
        self._seed_dev = None
        #: how many steps the HOST may run ahead of the device (0 = unbounded).  With the taped regions a step is enqueued in about
        #: half the time the device needs for it; unbounded, the host piles several steps of launches into the four hardware queues
        #: the ~14 streams share and blocks inside the runtime at unpredictable points (measured: 20.5 ms / step against 17.5 with
        #: the slower eager host, same GPU-only time).  Waiting for the END OF THE GENERATOR PHASE of an earlier step (one event per
        #: step, no device idle: the device still has >= one step queued) keeps the queues short.
        self.max_steps_ahead = int(os.environ.get("OSP_MAX_STEPS_AHEAD", "2"))
        self._pace = []
        self._g_done_event = None
        self._dstream = None
        self._disc_param_list = None
        self._reducers = None
        self.last_logs = {}

    # ------------------------------------------------------------------------------------------ plumbing
    @property
    def device(self):
        return next(self.parameters()).device

    def configure_optimizers(self):
        """base_lightning_module.py:47-71: two optimisers from the same partial, cosine-with-warm-up per step."""
        opt_f = self.hparams.optimizer or partial(FusedAdamW, lr=2e-4, betas=(0.8, 0.99), weight_decay=1e-2)
        opt_gen = opt_f([{"params": list(self.generator.parameters())}])
        opt_disc = opt_f([{"params": list(self.discriminator.parameters())}])
        max_steps = self.max_steps // 2
        sch_f = self.hparams.scheduler or partial(CosineWarmupSchedule, num_warmup_steps=1000)
        if isinstance(sch_f, partial) and "num_training_steps" in sch_f.keywords:
            sch_f.keywords["num_training_steps"] = max_steps
            sg, sd = sch_f(opt_gen, last_epoch=-1), sch_f(opt_disc, last_epoch=-1)
        else:
            sg = sch_f(opt_gen, num_training_steps=max_steps, last_epoch=-1)
            sd = sch_f(opt_disc, num_training_steps=max_steps, last_epoch=-1)
        return ([opt_gen, opt_disc], [{"scheduler": sg, "interval": "step"}, {"scheduler": sd, "interval": "step"}])

    def optimizers(self):
        if self._opts is None:
            opts, scheds = self.configure_optimizers()
            self._opts = (opts, [s["scheduler"] for s in scheds])
            self._reducers = (GradReducer(), GradReducer())
            for o, r in zip(opts, self._reducers):
                if hasattr(o, "arena"):
                    o.arena.reducer = r
            # data parallelism keeps replicas identical by construction from here on (same averaged gradients, same update);
            # the starting point is rank 0's weights and buffers, whatever each rank seeded or loaded
            self._reducers[0].broadcast_from_rank0([o.arena.data for o in opts] + [b for b in self.buffers()])
        return self._opts[0]

    def lr_schedulers(self):
        self.optimizers()
        return self._opts[1]

    # ------------------------------------------------------------------------------------------ training
    def _process_batch(self, batch):
        """base_lightning_module.py:24-45; the ground-truth segment gather happens on the device."""
        dev = self.device
        t = lambda v: v.to(dev, non_blocking=True) if v is not None else None      # noqa: E731
        from .. import precision as _prec
        with (_prec.parity_forward() if self.training else contextlib.nullcontext()):
            if (self.graph_segments and dev.type == "cuda" and self.training and torch.is_grad_enabled() and batch.get("sids") is None
                    and batch.get("lids") is None and self.train_args.gradient_accumulate_batches is None):
                gen_outputs = self._graphed_generator(tuple(t(batch[k]) for k in ("x", "x_lengths", "mel", "mel_lengths", "pitches", "energies")))
            elif (self.tape_segments and dev.type == "cuda" and self.training and torch.is_grad_enabled() and batch.get("sids") is None
                    and batch.get("lids") is None and rng.device_seed_active() and _tape.available()):
                gen_outputs = self._taped_generator(tuple(t(batch[k]) for k in ("x", "x_lengths", "mel", "mel_lengths", "pitches", "energies")))
            else:
                gen_outputs = self.generator(x=t(batch["x"]), x_lengths=t(batch["x_lengths"]), mel=t(batch["mel"]),
                                             mel_lengths=t(batch["mel_lengths"]), pitches=t(batch["pitches"]),
                                             energies=t(batch["energies"]), sids=t(batch.get("sids")), lids=t(batch.get("lids")))
        wav = batch["wav"]
        if isinstance(wav, np.ndarray):
            wav = torch.from_numpy(wav)
        wav = wav.to(dev, dtype=torch.float32, non_blocking=True)
        B, Tw = wav.shape
        hop, seg = self.hop_length, gen_outputs["segment_size"]
        if Tw % hop:
            wav = torch.nn.functional.pad(wav, (0, hop - Tw % hop))
        rows = wav.contiguous().view(B, -1, hop)
        gen_outputs["wav"] = K.gather_rows(rows, gen_outputs["start_idx"], seg).view(B, seg * hop)
        return gen_outputs

    def _taped_generator(self, tensors):
        """generator.forward through two taped segments (optispeech_amd/tape.py): the acoustic model with its losses, and the vocoder
        on its own stream.  The first step with a batch signature runs and records them; later steps replay."""
        from .. import precision
        gen = self.generator
        if self._tape_am is None:
            def am(x, x_lengths, mel, mel_lengths, pitches, energies, r01):
                o = gen._forward_am(x, x_lengths, mel, mel_lengths, pitches, energies, None, None, vocoder_hook=None, rand01=r01)
                a = o["_aux"]
                return (o["loss"], o["align_loss"], o["duration_loss"], o["pitch_loss"], o["energy_loss"], a["segment"], o["start_idx"],
                        a["durations"], a["p_avg"], a["e_avg"])
            self._tape_am = _tape.Segment(am, "acoustic model")
            self._tape_voc = _tape.Segment(lambda seg: (gen.vocoder(seg, f0=None),), "vocoder")
        x, mel = tensors[0], tensors[2]
        r01 = gen.draw_segment_rand(x.shape[0], x.device)
        key = (tuple((tuple(v.shape), v.dtype) for v in tensors), precision.signature(), id(self.optimizers()[0].arena))
        am = self._tape_am if os.environ.get("OSP_TAPE_AM", "1") != "0" else (lambda k, *a: self._tape_am.fn(*a))
        vo = self._tape_voc if os.environ.get("OSP_TAPE_VOC", "1") != "0" else (lambda k, *a: self._tape_voc.fn(*a))
        loss, align, dur, pit, ene, segment, start_idx, durations, p_avg, e_avg = am(key, *tensors, r01)
        seg_size = int(segment.shape[1])
        vkey = (tuple(segment.shape), precision.signature(), id(self.optimizers()[0].arena))
        from ..model.generator import _VOC_STREAM
        if _VOC_STREAM:
            wav_hat = ops.run_on_side_stream("vocoder", lambda: vo(vkey, segment)[0], [segment])
        else:
            wav_hat = vo(vkey, segment)[0]
        return {"wav_hat": wav_hat, "start_idx": start_idx, "segment_size": seg_size, "loss": loss,
                "align_loss": align.detach(), "duration_loss": dur.detach(), "pitch_loss": pit.detach(), "energy_loss": ene.detach(),
                "_aux": {"durations": durations, "p_avg": p_avg, "e_avg": e_avg, "segment": segment}}

    def _push_seed(self):
        """The step's dropout seed into device memory (the kernels of a taped region read it through ``seed_dev``: an argument
        baked into a recorded call would freeze the first step's masks)."""
        dev = self.device
        if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return False
        if self._seed_dev is None or self._seed_dev.device != dev:
            self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        K.store_i64(self._seed_dev, rng.host_seed())
        rng.use_device_seed(self._seed_dev)
        return True

    def _graphed_generator(self, tensors):
        """generator.forward through the two graphed segments (captured on the first call with this batch signature)."""
        from .. import ops, precision
        from ..graphs import GeneratorSegments
        key = (tuple((tuple(v.shape), str(v.dtype)) for v in tensors), precision.signature(), self.generator.segment_rand01 is not None)
        segs = self._gen_segments.get(key)
        if segs is None:
            if len(self._gen_segments) >= 4:
                self._gen_segments.pop(next(iter(self._gen_segments)))
            self.optimizers()                                  # the arenas must exist: captured kernels write the gradient arena
            segs = self._gen_segments[key] = GeneratorSegments(self, tensors)
        segs.push_seed()
        loss, align, dur, pit, ene, segment, start_idx = segs.am(*tensors)
        wav_hat = ops.run_on_side_stream("vocoder", lambda: segs.voc(segment)[0], [segment])
        return {"wav_hat": wav_hat, "start_idx": start_idx, "segment_size": segs.segment_size, "loss": loss,
                "align_loss": align.detach(), "duration_loss": dur.detach(), "pitch_loss": pit.detach(), "energy_loss": ene.detach()}

    def training_step(self, batch, batch_idx=0, **kwargs):
        """base_lightning_module.py:78-126.

        The step is a sequence of five stages (generator forward / generator backward / discriminator phase / two optimiser
        updates, ``_stage_*`` below) with the gradient all-reduces between them.  Eager mode issues them in the order below;
        with ``graph_steps`` the same stage functions are captured once into hipGraphs and replayed (``optispeech_amd/graphs.py``),
        which removes the ~20 ms of Python / autograd / dispatch time per step that bound the eager step."""
        ta = self.train_args
        accum = ta.gradient_accumulate_batches
        if (self.graph_steps and accum is None and torch.is_tensor(batch.get("x")) and self.device.type == "cuda"
                and ta.cache_generator_outputs):
            from ..graphs import graphed_training_step
            return graphed_training_step(self, batch, batch_idx)
        st = self._new_step_state(batch_idx)
        red_g, red_d = self._reducers
        # gradients are exchanged once per optimiser step: on the batches that only accumulate, nothing is all-reduced
        red_g.eager_ranges = red_d.eager_ranges = st.apply
        rng.advance()
        pace = self.max_steps_ahead > 0 and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing()
        if pace and len(self._pace) >= self.max_steps_ahead:
            self._pace.pop(0).synchronize()
        dev_seed = self.tape_segments and _tape.available() and self._push_seed()
        try:
            self._training_step_body(st, batch, red_g, red_d)
        finally:
            if dev_seed:
                rng.use_device_seed(None)
        if pace:
            ev = torch.cuda.Event()
            ev.record()                                            # end of this step's work on the calling stream
            self._pace.append(ev)

    def _training_step_body(self, st, batch, red_g, red_d):
        ta = self.train_args
        # ---- generator phase (discriminator weights frozen = toggle_optimizer; training_step_g freezes them after the
        # shared real-wave pass, which needs the parameter graph for the discriminator phase)
        self._stage_g_forward(st, batch)
        self._stage_g_backward(st)
        if st.apply:
            red_g.start_rest(self.optimizers()[0].arena.grad)
        # ---- discriminator phase (independent of the G update, so it overlaps the G-gradient all-reduce).
        # With ``pipeline_steps`` its loss / backward / optimizer step are issued from a second "calling" stream: the
        # calling stream proper only carries the generator work, so the NEXT step's generator forward (which needs the
        # updated generator weights, not the discriminator's) overlaps this step's discriminator backward.  The
        # discriminator stream is joined before anything reads discriminator state again (training_step_g, fetch_logs,
        # state_dict, join()).
        recompute = st.train_d and not ta.cache_generator_outputs
        opt_first = (not recompute) and st.train_d and self.pipeline_steps and _G_OPT_FIRST
        if recompute or opt_first:
            # cache_generator_outputs: false (base_lightning_module.py:111-113, :165-169): the discriminator phase re-runs the
            # generator without a tape -- AFTER the generator update (:103), so that update is issued first and the phase
            # cannot be pipelined behind the next step's generator forward
            red_g.wait()
            self._stage_opt_g(st)
        dctx = self._disc_phase_stream() if (st.train_d and self.pipeline_steps and not recompute) else contextlib.nullcontext()
        if st.train_d:
            with dctx:
                self._stage_d(st, batch)
                # every sub-discriminator's slice was already launched when its backward finished (dp.reduce_ready, from
                # disc_ops._stack_backward): gradient-ready order, overlapping the other stacks' backward; this adds the rest
                if st.apply:
                    red_d.start_rest(self.optimizers()[1].arena.grad)
        if not (recompute or opt_first):
            red_g.wait()
            self._stage_opt_g(st)
        if st.train_d:
            with dctx:
                red_d.wait()
                self._stage_opt_d(st)
        self.last_logs = st.logs

    # The five stages of a step.  ``st`` carries what flows between them; every stage only enqueues device work.
    def _new_step_state(self, batch_idx=0):
        ta = self.train_args
        accum = ta.gradient_accumulate_batches
        self.optimizers()
        return SimpleNamespace(scale=float(accum) if accum is not None else 1.0,
                               apply=((batch_idx + 1) % accum == 0) if accum is not None else True,
                               train_d=self.global_step >= ta.pretraining_steps, logs={}, loss_g=None, wav=None, wav_hat=None,
                               pre=None)

    def _stage_g_forward(self, st, batch):
        cached = self.train_args.cache_generator_outputs      # (false: the discriminator phase draws its own segment)
        st.loss_g, (st.wav, st.wav_hat) = self.training_step_g(batch, st.train_d, st.logs,
                                                               share_real=st.train_d and self.share_real_pass and cached)
        # the discriminator-phase inputs are fixed from here on: stage them before the generator's backward is queued, so
        # that the discriminator-phase forward (side streams) overlaps that backward (this stream)
        st.pre = (self.discriminator.prepare_disc_inputs(st.wav, st.wav_hat.detach())
                  if st.train_d and self._real_pass is None and cached else None)

    def _stage_g_backward(self, st):
        if st.apply:
            self.optimizers()[0].zero_grad()
        ops.begin_backward()
        with torch.autograd.set_multithreading_enabled(_AUTOGRAD_MT):
            (st.loss_g / st.scale).backward()
        st.loss_g = None
        self._g_done_event = torch.cuda.current_stream().record_event() if _D_AFTER_G else None
        if _SYNC_AFTER_G:                                  # diagnostic (tools/race2.sh (git history)): drain the device between the phases
            torch.cuda.synchronize()

    def _stage_d(self, st, batch):
        for p in self._disc_params():
            p.requires_grad_(True)
        if not self.train_args.cache_generator_outputs:
            rng.advance()                                      # its own dropout / DropPath masks, as a second forward has
            with torch.no_grad():                              # :165-169 -- a fresh forward (new segment draw, updated generator)
                again = self._process_batch(batch)
            st.wav, st.wav_hat, st.pre = again["wav"], again["wav_hat"], None
            self._real_pass = None
        # share_real_pass: two stack nodes (the kept real pass + the generated branch) accumulate into each
        # sub-discriminator's arena slice, so a slice is only complete when BOTH backwards ran -- no gradient-ready
        # collectives from inside this backward; the closing start_rest covers the whole arena
        red_d = self._reducers[1] if self._reducers is not None else None
        shared = self._real_pass is not None
        keep_ranges = getattr(red_d, "eager_ranges", True)
        if shared and red_d is not None:
            red_d.eager_ranges = False
        if _D_AFTER_G and st.pre is not None:
            # experiment knob: the discriminator phase's forward starts only after the generator's backward has drained from the
            # calling stream (instead of as soon as the waves exist)
            st.pre = (st.pre[0], self._g_done_event if self._g_done_event is not None else torch.cuda.current_stream().record_event())
        try:
            replay = self.replay_disc_forward and self.train_args.cache_generator_outputs
            inline = None
            if (_INLINE_D and self.pipeline_steps and st.apply and not shared and not replay and st.pre is not None
                    and not torch.cuda.is_current_stream_capturing()):
                # every sub-discriminator goes forward -> hinge -> backward on its own stream without the phase-wide join in
                # between (model/discriminator.py: _forward_concurrent); the arena is cleared first, and the stacks wait for that
                ev, self._d_zeroed = getattr(self, "_d_zeroed", None), None
                if ev is None:
                    self.optimizers()[1].zero_grad()
                    ev = torch.cuda.current_stream().record_event()
                inline = (1.0 / st.scale, ev)
            loss_d = self.training_step_d(batch, (st.wav, st.wav_hat.detach()), st.logs, pre=st.pre, replay=replay, inline_backward=inline)
            if loss_d.requires_grad:
                if st.apply and inline is None:
                    self.optimizers()[1].zero_grad()
                elif inline is not None:
                    raise RuntimeError("inline discriminator backward was requested but the phase returned a differentiable loss")
                ops.begin_backward()
                with torch.autograd.set_multithreading_enabled(_AUTOGRAD_MT):
                    (loss_d / st.scale).backward()
        finally:
            if shared and red_d is not None:
                red_d.eager_ranges = keep_ranges
        st.pre = None

    def _stage_opt_g(self, st):
        if st.apply:
            self.optimizers()[0].step(max_norm=self.train_args.gradient_clip_val, grad_scale=1.0 / self._reducers[0].world)
            self.lr_schedulers()[0].step()
            self.global_step += 1

    def _stage_opt_d(self, st):
        if st.apply:
            self.optimizers()[1].step(max_norm=self.train_args.gradient_clip_val, grad_scale=1.0 / self._reducers[1].world)
            self.lr_schedulers()[1].step()
            self.global_step += 1
            if _INLINE_D and self.pipeline_steps and not torch.cuda.is_current_stream_capturing():
                # inline discriminator backward: clear the arena NOW (nothing writes it before the next discriminator phase: the
                # generator phase runs with the discriminator frozen), so that the next phase's stacks need not wait for a clear that
                # is ordered behind the generator's whole backward
                self.optimizers()[1].zero_grad()
                self._d_zeroed = torch.cuda.current_stream().record_event()

    def _disc_params(self):
        """The discriminator's parameter list, walked once (toggled twice per step: toggle_optimizer of the reference)."""
        if self._disc_param_list is None:
            self._disc_param_list = list(self.discriminator.parameters())
        return self._disc_param_list

    def _disc_phase_stream(self):
        """Context manager: make the discriminator-phase stream current, ordered after the calling stream's work so far."""
        if self._dstream is None:
            from .. import lanes
            self._dstream = lanes.stream("dphase", self.device) if lanes.managed() else torch.cuda.Stream(device=self.device)
        ev, self._dgrad_done = getattr(self, "_dgrad_done", None), None
        if ev is not None:
            self._dstream.wait_event(ev)
        else:
            self._dstream.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self._dstream)

    def join(self):
        """The current stream waits for a discriminator phase still in flight (``pipeline_steps``); no host sync."""
        if self._dstream is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_stream(self._dstream)

    def state_dict(self, *args, **kwargs):
        self.join()
        return super().state_dict(*args, **kwargs)

    def training_step_g(self, batch, train_discriminator, logs, share_real=False):
        """base_lightning_module.py:128-161 (log values stay on the device; see fetch_logs).

        share_real: evaluate the discriminators on the real waves once, with the parameter graph, and keep the result
        for training_step_d of the same step (same weights, same waves -> same values as the reference's two passes)."""
        gen_outputs = self._process_batch(batch)
        gen_am_loss = gen_outputs["loss"]
        logs.update({"total_loss/train_am_loss": gen_am_loss.detach(),
                     "gen_subloss/train_alighn_loss": gen_outputs["align_loss"],
                     "gen_subloss/train_duration_loss": gen_outputs["duration_loss"],
                     "gen_subloss/train_pitch_loss": gen_outputs["pitch_loss"],
                     "gen_subloss/train_energy_loss": gen_outputs["energy_loss"]})
        wav, wav_hat = gen_outputs["wav"], gen_outputs["wav_hat"]
        self._real_pass = None
        self._dgrad_done = None
        if train_discriminator and self.pipeline_steps and _EARLY_D and wav_hat.requires_grad and wav_hat.is_cuda:
            # The discriminator phase reads the discriminator weights and wav_hat.detach(): nothing of the generator's backward
            # BELOW the discriminators (vocoder, acoustic model: ~3 ms of small-grid kernels).  It may start as soon as the generator
            # phase has left the discriminator stacks, i.e. when the gradient w.r.t. wav_hat is complete -- this hook fires then
            # (all stack streams joined into the calling stream), and the discriminator-phase stream waits for THIS event instead
            # of the end of the generator's backward (profiles/r05_phase_events_*.txt)
            def _mark(grad, self=self):
                if not torch.cuda.is_current_stream_capturing():
                    self._dgrad_done = torch.cuda.current_stream().record_event()
                return None
            wav_hat.register_hook(_mark)
        if train_discriminator:
            self.join()                          # a pipelined discriminator update of the previous step must have landed
            if share_real:
                self._real_pass = self.discriminator.forward_real(wav)
            for p in self._disc_params():
                p.requires_grad_(False)
            gen_adv_loss, log_dict = self.discriminator.forward_gen(wav, wav_hat, real=self._real_pass)
            logs["total_loss/train_gen_adv_loss"] = gen_adv_loss.detach()
            logs.update({f"gen_adv_loss/train_{k}": v for k, v in log_dict.items()})
            loss = gen_am_loss + gen_adv_loss
        else:
            loss = gen_am_loss
        logs["total_loss/generator"] = loss.detach()
        self._last_gen_outputs = gen_outputs
        return loss, (wav.detach(), wav_hat)

    def training_step_d(self, batch, wav_outputs, logs, pre=None, replay=False, inline_backward=None):
        """base_lightning_module.py:163-186; D sees wav_hat.detach()."""
        wav, wav_hat = wav_outputs
        real, self._real_pass = getattr(self, "_real_pass", None), None
        loss, log_dict = self.discriminator.forward_disc(wav, wav_hat, real=real, pre=pre, replay=replay, inline_backward=inline_backward)
        logs["total_loss/discriminator"] = loss.detach()
        logs.update({f"discriminator/{k}": v for k, v in log_dict.items()})
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0, **kwargs):
        """base_lightning_module.py:195-254 without the third-party perceptual scores (periodicity / UTMOS / PESQ are outside the
        hot path: their terms are the zeros the reference uses when the evaluate_* flags are off).  Returns the logged dict:
        total_loss/val_am_loss, gen_subloss/val_*, total_loss/val_gen_adv_loss, gen_adv_loss/val_{mel_loss,mr_stft_loss},
        total_loss/val_total -- one device->host copy for all of them."""
        self.join()
        gen_outputs = self._process_batch(batch)
        wav, wav_hat = gen_outputs["wav"], gen_outputs["wav_hat"]
        gen_adv_loss, log_dict = self.discriminator.forward_val(wav, wav_hat)
        logs = {"total_loss/val_am_loss": gen_outputs["loss"], "gen_subloss/val_alighn_loss": gen_outputs["align_loss"],
                "gen_subloss/val_duration_loss": gen_outputs["duration_loss"], "gen_subloss/val_pitch_loss": gen_outputs["pitch_loss"],
                "gen_subloss/val_energy_loss": gen_outputs["energy_loss"], "total_loss/val_gen_adv_loss": gen_adv_loss}
        logs.update({f"gen_adv_loss/val_{k}": v for k, v in log_dict.items()})
        logs["total_loss/val_total"] = gen_outputs["loss"] + gen_adv_loss
        keys = list(logs)
        packed = torch.stack([logs[k].detach().float().reshape(()) for k in keys])
        if self._reducers is not None:
            self._reducers[0].mean_scalars(packed)
        return dict(zip(keys, packed.cpu().tolist()))

    def fetch_logs(self):
        """All logged scalars with ONE device->host copy (and one packed all-reduce under data parallelism)."""
        if not self.last_logs:
            return {}
        self.join()
        keys = list(self.last_logs)
        packed = torch.stack([self.last_logs[k].float().reshape(()) for k in keys])
        if self._reducers is not None:
            self._reducers[0].mean_scalars(packed)
        vals = packed.cpu().tolist()
        return dict(zip(keys, vals))

    # ------------------------------------------------------------------------------------------ inference
    @torch.inference_mode()
    def synthesise(self, inputs: InferenceInputs, durations_override=None) -> InferenceOutputs:
        """optispeech.py:58-81."""
        inputs = inputs.as_torch().to(self.device)
        out = self.generator.synthesise(x=inputs.x, x_lengths=inputs.x_lengths.to("cpu"), sids=inputs.sids,
                                        lids=inputs.lids, d_factor=inputs.d_factor, p_factor=inputs.p_factor,
                                        e_factor=inputs.e_factor, durations_override=durations_override)
        return InferenceOutputs(wav=out["wav"], wav_lengths=out["wav_lengths"], durations=out["durations"],
                                pitch=out["pitch"], energy=out["energy"], latency=out["latency"], rtf=out["rtf"],
                                am_rtf=out["am_rtf"], v_rtf=out["v_rtf"])

    synthesize = synthesise            # README spelling (README.md:90)

    def prepare_input(self, text, *, language=None, speaker=None, d_factor=None, p_factor=None, e_factor=None,
                      split_sentences=True) -> InferenceInputs:
        """optispeech.py:83-154."""
        languages = self.text_processor.languages
        if language is None:
            language = languages[0]
        if self.num_speakers > 1:
            if speaker is None:
                sid = 0
            elif type(speaker) is str:
                try:
                    sid = self.speakers.index(speaker)
                except (IndexError, ValueError, AttributeError):
                    raise ValueError(f"A speaker with the given name `{speaker}` was not found in speaker list")
            else:
                sid = int(speaker)
        else:
            sid = None
        if self.text_processor.is_multi_language:
            try:
                lid = languages.index(language)
            except (IndexError, ValueError):
                raise ValueError(f"A language with the given name `{language}` was not found in language list")
        else:
            lid = None
        input_ids, clean_text = self.text_processor(text, lang=language, split_sentences=split_sentences)
        if split_sentences:
            lengths = [len(p) for p in input_ids]
        else:
            lengths, input_ids = [len(input_ids)], [input_ids]
        sids = [sid] * len(input_ids) if sid is not None else None
        lids = [lid] * len(input_ids) if lid is not None else None
        ia = self.inference_args
        inputs = InferenceInputs.from_ids_and_lengths(ids=input_ids, lengths=lengths, clean_text=clean_text, sids=sids,
                                                      lids=lids, d_factor=d_factor or ia.d_factor,
                                                      p_factor=p_factor or ia.p_factor, e_factor=e_factor or ia.e_factor)
        return inputs.as_torch().to(self.device)

    # ------------------------------------------------------------------------------------------ ONNX-compatible I/O
    @torch.inference_mode()
    def onnx_io(self, x, x_lengths, scales, sids=None, lids=None):
        """The signature of the reference's exported graph (onnx/export.py:40-81): inputs ``x`` (B, T) int64,
        ``x_lengths`` (B,), ``scales`` = [d_factor, p_factor, e_factor] -> (wav, wav_lengths, durations), so consumers of
        the ``ospeech`` runtime (onnx/infer.py:24-145) can call the native model with the tensors they already build."""
        sc = [float(v) for v in (scales.tolist() if hasattr(scales, "tolist") else scales)]
        dev = self.device
        as_t = lambda v, dt: torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(dtype=dt)   # noqa: E731
        out = self.generator.synthesise(x=as_t(x, torch.long).to(dev), x_lengths=as_t(x_lengths, torch.long).to("cpu"),
                                        sids=as_t(sids, torch.long).to(dev) if sids is not None else None,
                                        lids=as_t(lids, torch.long).to(dev) if lids is not None else None,
                                        d_factor=sc[0], p_factor=sc[1], e_factor=sc[2])
        return out["wav"], out["wav_lengths"], out["durations"]

    # ------------------------------------------------------------------------------------------ checkpoints
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, config=None, strict=False, **kwargs):
        """Read a reference (Lightning) ``.ckpt``: only ``state_dict`` is used; the pickled Hydra partials /
        OmegaConf nodes of ``hyper_parameters`` are skipped by a tolerant unpickler, and
        the architecture comes from ``config`` (default: the BASELINE ConvNeXt configuration)."""
        from ..config import ModelConfig, make_optispeech
        ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", pickle_module=_TolerantPickle,
                          weights_only=False)
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        model = make_optispeech(config or ModelConfig())
        missing, unexpected = model.load_state_dict(sd, strict=False)
        bad = [k for k in missing if "melspec_loss" not in k and "window" not in k]
        if strict and (bad or unexpected):
            raise RuntimeError(f"checkpoint mismatch: missing {bad[:5]}, unexpected {list(unexpected)[:5]}")
        model.ckpt_loaded_epoch = ckpt.get("epoch") if isinstance(ckpt, dict) else None    # on_load_checkpoint :305-306
        return model


    def _moment_views(self, opt):
        """(reference key, to_ref, to_native, exp_avg view, exp_avg_sq view) for every parameter of a FusedAdamW arena."""
        by_id = {}
        for mprefix, mod in self.named_modules():
            for name, prm in mod._parameters.items():
                if prm is None:
                    continue
                key, to_native, to_ref = mod._ref(name) if hasattr(mod, "_ref") else (name, None, None)
                by_id[id(prm)] = ((mprefix + "." if mprefix else "") + key, to_ref, to_native)
        for prm, off in zip(opt.arena.params, opt.arena.offsets):
            key, to_ref, to_native = by_id[id(prm)]
            n = prm.numel()
            yield key, to_ref, to_native, opt.exp_avg[off:off + n].view(prm.shape), opt.exp_avg_sq[off:off + n].view(prm.shape)

    def save_checkpoint(self, path):
        """Write a checkpoint the reference can read: ``state_dict`` in the reference's key / layout schema (weight-norm
        g / v, conv kernels (Cout, Cin, k), un-padded head) plus ``epoch`` / ``global_step``
        (base_lightning_module.py:305-306 reads ``epoch``).  The native extras (AdamW moments keyed by reference
        parameter name and stored in the reference layout, step counters, schedule positions, dropout RNG position)
        live under ``"osp"`` so that a resumed run continues exactly."""
        from .. import rng
        self.join()                               # a pipelined discriminator update may still be writing weights / moments
        opts = self.optimizers()
        extra = {"rng": dict(rng._state), "global_step": self.global_step, "optimizers": []}
        for opt, sch in zip(opts, self.lr_schedulers()):
            mom = {k: ((to_ref(a) if to_ref else a).detach().cpu().clone(), (to_ref(b) if to_ref else b).detach().cpu().clone())
                   for k, to_ref, _, a, b in self._moment_views(opt)}
            extra["optimizers"].append({"step": opt.step_count, "lr": opt.lr, "last_step": sch.last_step, "moments": mom})
        torch.save({"state_dict": {k: v.detach().cpu().clone() for k, v in self.state_dict().items()},
                    "epoch": getattr(self, "ckpt_loaded_epoch", None) or 0, "global_step": self.global_step, "osp": extra}, path)

    def load_training_state(self, ckpt):
        """Restore what save_checkpoint put under ``"osp"`` (after the weights were loaded and the model moved to its device)."""
        from .. import rng
        extra = ckpt["osp"] if "osp" in ckpt else ckpt
        self.join()
        self._step_graphs.clear()                 # captured steps hold the old optimiser scalars
        rng._state.update(extra["rng"])
        self.global_step = int(extra["global_step"])
        for opt, sch, st in zip(self.optimizers(), self.lr_schedulers(), extra["optimizers"]):
            opt.step_count, opt.lr, sch.last_step = int(st["step"]), float(st["lr"]), int(st["last_step"])
            for key, _, to_native, a, b in self._moment_views(opt):
                ea, eb = st["moments"][key]
                a.copy_((to_native(ea) if to_native else ea).to(a.device))
                b.copy_((to_native(eb) if to_native else eb).to(b.device))


    def load_lightning_training_state(self, ckpt):
        """Optimizer / schedule state of a reference (Lightning) ``.ckpt`` -> the fused AdamW arenas: ``optimizer_states`` (one
        ``torch.optim.AdamW.state_dict()`` per optimizer, base_lightning_module.py:47-71: [generator, discriminator]) stores
        ``exp_avg`` / ``exp_avg_sq`` / ``step`` by parameter INDEX in ``module.parameters()`` order, which is the order of the
        checkpoint's own ``state_dict`` keys restricted to the parameters (pinned by tests/golden/ref_param_order.npz, produced from
        the reference modules).  Moments are converted to the kernel-native layouts like the weights.  ``ckpt``: path or loaded dict.
        Call after ``load_from_checkpoint(...).to(device)``."""
        if not isinstance(ckpt, dict):
            ckpt = torch.load(ckpt, map_location="cpu", pickle_module=_TolerantPickle, weights_only=False)
        self.join()
        self._step_graphs.clear()
        sd_keys = list(ckpt["state_dict"].keys())
        scheds = ckpt.get("lr_schedulers") or [None, None]
        for i, (opt, sch, prefix) in enumerate(zip(self.optimizers(), self.lr_schedulers(), ("generator.", "discriminator."))):
            ost = ckpt["optimizer_states"][i]
            views = {key: (to_native, a, b) for key, _, to_native, a, b in self._moment_views(opt)}
            pkeys = [k for k in sd_keys if k.startswith(prefix) and k in views]
            ids = [pid for grp in ost["param_groups"] for pid in grp["params"]]
            if len(ids) != len(pkeys) or len(pkeys) != len(views):
                raise RuntimeError(f"optimizer {i}: checkpoint lists {len(ids)} parameters, its state_dict has {len(pkeys)} of this "
                                   f"model's {len(views)} under {prefix!r}")
            step = 0
            for pid, key in zip(ids, pkeys):
                st = ost["state"].get(pid)
                to_native, a, b = views[key]
                if st is None:                                # a parameter that never received a gradient
                    a.zero_(); b.zero_()
                    continue
                ea, eb = st["exp_avg"].float(), st["exp_avg_sq"].float()
                a.copy_((to_native(ea) if to_native else ea).to(a.device))
                b.copy_((to_native(eb) if to_native else eb).to(b.device))
                step = max(step, int(st["step"]))
            opt.step_count = step
            opt.lr = float(ost["param_groups"][0]["lr"])
            if scheds[i] is not None and "last_epoch" in scheds[i]:
                sch.last_step = int(scheds[i]["last_epoch"])
        if "global_step" in ckpt:
            self.global_step = int(ckpt["global_step"])


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __setstate__(self, state):
        pass


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            return _Dummy


class _TolerantPickle:
    """pickle-module shim for torch.load: unknown classes (lightning / omegaconf / hydra / optispeech.*) -> inert."""
    __name__ = "tolerant_pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _TolerantUnpickler(io.BytesIO(b), **kw).load())
