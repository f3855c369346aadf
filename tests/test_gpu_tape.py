"""Taped regions of the training step (optispeech_amd/tape.py) against their eager execution ("a -m gpu test
that the replayed step equals the eager step").

The kernels a replay launches are the ones the recording run launched -- same entry points, same arguments up to the patched input
addresses, same streams -- so everything deterministic must agree BIT FOR BIT: forward activations, scores, input gradients.  The
weight-gradient kernels accumulate with f32 atomics (split-K): their sums differ in the last bits between ANY two runs, eager or
not, and are compared at 2e-5 of the tensor's scale.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stack_run(make, waves_per_iter, tapes):
    """Three forward + backward passes of one sub-discriminator (fresh input allocations, weights changed between the passes as an
    optimizer would) -> per pass (scores, feature maps, d waves, parameter gradients)."""
    from optispeech_amd import precision, tape, values
    from optispeech_amd.ops import gsink
    keep = tape.ENABLED
    tape.ENABLED = tapes
    precision.set_precision("bf16")
    try:
        torch.manual_seed(0)
        d = make().to(DEV)
        out = []
        for it, wav in enumerate(waves_per_iter):
            x = wav.clone().to(DEV).requires_grad_(True)
            junk = torch.empty((it + 1) * 1000 + 17, device=DEV)          # shifts the allocator: inputs live elsewhere each pass
            for p in d.parameters():                                      # zeroed IN PLACE: the gradient sinks keep their addresses, as
                gsink(p).zero_()                                          # the optimizer's flat arena does (a tape holds them)
            for half in (0, x.shape[0] // 2):                             # whole-batch pass and the generator phase's no-grad head
                for p in d.parameters():
                    p.requires_grad_(half == 0)
                res = d(x, nograd_head=half) if half else d(x)
                (score, fmaps) = res[1] if half else res
                g = torch.Generator(device="cpu").manual_seed(100 + it)
                loss = (score * torch.randn(score.shape, generator=g).to(DEV)).sum()
                for f in fmaps[:-1]:
                    loss = loss + (f.float() * torch.randn(f.shape, generator=g).to(DEV)).mean()
                x.grad = None
                loss.backward()
                torch.cuda.synchronize()
                grads = {k: gsink(p).detach().clone() for k, p in d.named_parameters()} if half == 0 else {}
                out.append((score.detach().clone(), [f.detach().float().clone() for f in fmaps], x.grad.detach().clone(), grads))
            del junk
            with torch.no_grad():                                         # "optimizer step": new weights, stale packs
                for p in d.parameters():
                    p.add_(torch.randn(p.shape, generator=torch.Generator().manual_seed(7 + it)).to(DEV) * 0.01 * p.abs().mean())
                    p.requires_grad_(True)
            values.bump_param_epoch()
        return out, tape.stats()
    finally:
        tape.ENABLED = keep
        precision.set_precision("f32")


@pytest.mark.parametrize("which", ["period3", "period11", "resolution"])
def test_taped_discriminator_stack_equals_eager(which):
    from optispeech_amd import tape
    from optispeech_amd.model.discriminator import DiscriminatorP, DiscriminatorR
    make = {"period3": lambda: DiscriminatorP(3), "period11": lambda: DiscriminatorP(11),
            "resolution": lambda: DiscriminatorR((1024, 256, 1024))}[which]
    g = torch.Generator().manual_seed(5)
    waves = [torch.rand(4, 16384, generator=g) * 2 - 1 for _ in range(3)]
    eager, _ = _stack_run(make, waves, tapes=False)
    s0 = tape.stats()
    taped, s1 = _stack_run(make, waves, tapes=True)
    assert tape.available()
    assert s1["recorded"] - s0["recorded"] >= 4, (s0, s1)               # forward + backward, two phases
    assert s1["replayed"] - s0["replayed"] >= 8, (s0, s1)               # passes 2 and 3 are replays
    assert s1["poisoned"] == s0["poisoned"], "a stack region fell back to eager execution"
    for i, (e, t) in enumerate(zip(eager, taped)):
        assert torch.equal(e[0], t[0]), f"pass {i}: scores differ"
        for a, b in zip(e[1], t[1]):
            assert torch.equal(a, b), f"pass {i}: feature maps differ"
        if which == "resolution":
            # the wave gradient of a resolution stack passes through the STFT backward, whose overlap-add accumulates with f32
            # atomics: not bit-reproducible between ANY two runs
            assert (e[2] - t[2]).abs().max().item() <= 1e-5 * e[2].abs().max().item(), f"pass {i}: input gradient differs"
        else:
            assert torch.equal(e[2], t[2]), f"pass {i}: input gradient differs"
        for k in e[3]:
            scale = e[3][k].abs().max().item()
            assert (e[3][k] - t[3][k]).abs().max().item() <= 2e-5 * scale + 1e-12, (i, k)


def _train(tapes, steps=4, pipeline=True, segments=False):
    from optispeech_amd import precision, rng, tape
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    keep = tape.ENABLED
    tape.ENABLED = tapes
    precision.set_precision("bf16")
    try:
        cfg = ModelConfig()                                               # BASELINE widths, dropout / drop-path ON
        batch = synthetic_batch(2, 24, 96, cfg, seed=5, device=DEV)
        torch.manual_seed(3)
        torch.cuda.manual_seed(3)
        rng.manual_seed(3, 0)
        rng._state["next_stream"] = 1
        m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to(DEV).train()
        m.pipeline_steps = pipeline
        m.tape_segments = segments
        m.generator.segment_rand01 = torch.rand(2, generator=torch.Generator().manual_seed(1)).to(DEV)
        m.optimizers()
        for sch in m.lr_schedulers():
            sch.warmup = 4
            sch.opt.lr = sch.base_lr * (1.0 / 4)
            sch.last_step = 1
        logs = []
        for i in range(steps):
            m.training_step(batch, i)
            logs.append(m.fetch_logs())
        m.join()
        torch.cuda.synchronize()
        return logs, {k: v.detach().clone() for k, v in m.state_dict().items()}, tape.stats()
    finally:
        tape.ENABLED = keep
        precision.set_precision("f32")


@pytest.mark.parametrize("backbone", ["convnext", "transformer"])
def test_taped_segments_on_changing_ragged_batches_equal_eager(backbone):
    """(transformer: the Transformer-backbone acoustic model is a recordable region since the scaled positional encoding, the head
    split / merge and the dropout sites are launches of ours -- ops.ScaledPosEncFn, osp_permute_0213, osp_dropout_add.)
    A taped generator segment replayed on ANOTHER ragged batch of the same padded shape (what real training does every step; the
    benchmark re-uses one batch and cannot see it).  Round 5 found the backward tape reading the recording step's token ids and
    lengths (they were not declared inputs of the backward region): 2.5e-2 wrong acoustic-model gradients.  Steps: batch 0
    (records), 1, 1, 0 without an optimizer update; gradients of both networks, taped vs eager, to the tolerance of two eager
    runs (f32 atomics)."""
    from oracle import schema as S
    from optispeech_amd import precision, rng, tape
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    precision.set_precision("f32")
    keep = tape.ENABLED

    def run(tapes):
        tape.ENABLED = tapes
        c = S.SMALL
        cfg = ModelConfig(dim=c.dim, enc_inter=c.enc_inter, dec_inter=c.dec_inter, dur=c.dur + (0.0,), pitch=c.pitch + (0.0,),
                          energy=c.energy + (0.0,), voc_dim=c.voc_dim, voc_inter=c.voc_inter, voc_layers=c.voc_layers,
                          backbone=backbone).no_dropout()
        torch.manual_seed(7); rng.manual_seed(7, 0)
        m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to(DEV).train()
        m.tape_segments = True
        got = {}
        for name, o in zip(("g", "d"), m.optimizers()):
            o.step = (lambda n, oo: (lambda *a, **k: got.__setitem__(n, oo.arena.grad.detach().clone())))(name, o)
        out = []
        for r in (0, 1, 1, 0):
            b = synthetic_batch(2, 24, 96, cfg, seed=50 + r, ragged=True, device=DEV)
            m.generator.segment_rand01 = torch.tensor([0.25 + 0.5 * r, 0.6 - 0.3 * r], device=DEV)
            m.training_step(b, 0)
            torch.cuda.synchronize()
            out.append((got["g"].clone(), got["d"].clone()))
        return out

    try:
        s0 = tape.stats()
        a, b = run(False), run(True)
        s1 = tape.stats()
    finally:
        tape.ENABLED = keep
    assert s1["replayed"] - s0["replayed"] >= 6 and s1["poisoned"] == s0["poisoned"], (s0, s1)
    for i, ((ga, da), (gb, db)) in enumerate(zip(a, b)):
        assert ((ga - gb).norm() / ga.norm()).item() < 1e-5, (i, "generator")
        assert ((da - db).norm() / da.norm()).item() < 1e-5, (i, "discriminator")


@pytest.mark.parametrize("mode", ["mixed", "bf16"])
def test_taped_segments_across_differently_shaped_micro_batches_see_fresh_weights(mode):
    """gradient_accumulate_batches = 2 with two padded shapes A, B.  In one optimizer epoch micro-batch A records
    its tapes (first user of the epoch: its tape carries the weight-pack refresh) and micro-batch B records ITS tapes when every
    pack is already fresh -- before the fix B's tapes held no refresh launch, and an epoch that begins with B (order B, A) replayed
    them on the previous epoch's weights: silently stale forward and input gradients.  The "optimizer" here moves the weights by a
    visible amount (1 % of each tensor's scale) so that a stale pack cannot hide inside the tolerance.  Order: (A, B) eager,
    (A, B) records, (B, A) and (B, A) replay; accumulated gradients of every epoch, taped vs eager."""
    from oracle import schema as S
    from optispeech_amd import precision, rng, tape, values
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    precision.set_precision(mode)
    keep = tape.ENABLED

    def run(tapes):
        tape.ENABLED = tapes
        c = S.SMALL
        cfg = ModelConfig(dim=c.dim, enc_inter=c.enc_inter, dec_inter=c.dec_inter, dur=c.dur + (0.0,), pitch=c.pitch + (0.0,),
                          energy=c.energy + (0.0,), voc_dim=c.voc_dim, voc_inter=c.voc_inter, voc_layers=c.voc_layers).no_dropout()
        torch.manual_seed(7); rng.manual_seed(7, 0)
        m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to(DEV).train()
        m.train_args.gradient_accumulate_batches = 2
        m.tape_segments = True
        m.pipeline_steps = False
        got, epoch = {}, [0]

        def fake_step(name, oo):
            def step(*a, **k):
                got[name] = oo.arena.grad.detach().clone()
                noise = torch.randn(oo.arena.numel, generator=torch.Generator().manual_seed(100 * epoch[0] + len(name))).to(DEV)
                oo.arena.data.add_(noise * 0.01 * oo.arena.data.abs().mean())
                values.bump_param_epoch()
            return step
        for name, o in zip(("g", "dd"), m.optimizers()):
            o.step = fake_step(name, o)
        shapes = {"A": (24, 96), "B": (20, 80)}
        out, idx = [], 0
        for pair in ("AB", "AB", "BA", "BA"):
            for which in pair:
                tt, tm = shapes[which]
                b = synthetic_batch(2, tt, tm, cfg, seed=60 + idx, ragged=True, device=DEV)
                m.generator.segment_rand01 = torch.tensor([0.25, 0.6], device=DEV)
                m.training_step(b, idx)
                idx += 1
            m.join()
            torch.cuda.synchronize()
            epoch[0] += 1
            out.append((got["g"].clone(), got["dd"].clone()))
        return out

    try:
        s0 = tape.stats()
        a = run(False)
        b = run(True)
        s1 = tape.stats()
    finally:
        tape.ENABLED = keep
        precision.set_precision("f32")
    assert s1["replayed"] - s0["replayed"] >= 8 and s1["poisoned"] == s0["poisoned"], (s0, s1)
    tol = 1e-5 if mode == "mixed" else 2e-3
    for i, ((ga, da), (gb, db)) in enumerate(zip(a, b)):
        assert ((ga - gb).norm() / ga.norm()).item() < tol, (i, "generator", ((ga - gb).norm() / ga.norm()).item())
        assert ((da - db).norm() / da.norm()).item() < tol, (i, "discriminator", ((da - db).norm() / da.norm()).item())


def test_two_forwards_before_either_backward_keep_their_activations():
    """a taped forward's outputs are the tape's buffers.  Two forwards of one stack with the same key before
    either backward (what OSP_SHARE_REAL=1 does: forward_real(wav), then d(wav_hat.detach())) must not share them -- the second
    forward takes another lease slot (disc_ops._Lease), i.e. its own buffer set.  Input gradients are deterministic: taped ==
    eager bit for bit; weight gradients (now atomic-free as well) too."""
    from optispeech_amd import precision, tape
    from optispeech_amd.model.discriminator import DiscriminatorP
    from optispeech_amd.ops import gsink
    precision.set_precision("bf16")
    keep = tape.ENABLED
    res = {}
    try:
        for mode in (False, True):
            tape.ENABLED = mode
            torch.manual_seed(0)
            d = DiscriminatorP(3).to(DEV)
            outs = []
            for it in range(3):                                          # pass 0 records (both slots), passes 1-2 replay
                g = torch.Generator().manual_seed(50 + it)
                xa = torch.randn(4, 8192, generator=g).to(DEV).requires_grad_(True)
                xb = torch.randn(4, 8192, generator=g).to(DEV).requires_grad_(True)
                for p in d.parameters():
                    gsink(p).zero_()
                sa, fa = d(xa)
                sb, fb = d(xb)                                           # same shapes, same weights, same key: would replay over sa / fa
                wa = torch.randn(sa.shape, generator=g).to(DEV)
                wb = torch.randn(sb.shape, generator=g).to(DEV)
                la = (sa * wa).sum() + sum((f.float() ** 2).mean() for f in fa[:-1])
                lb = (sb * wb).sum() + sum((f.float() ** 2).mean() for f in fb[:-1])
                la.backward()
                lb.backward()
                torch.cuda.synchronize()
                outs.append((xa.grad.clone(), xb.grad.clone(), sa.detach().clone(), {k: gsink(p).detach().clone() for k, p in d.named_parameters()}))
            res[mode] = outs
    finally:
        tape.ENABLED = keep
        precision.set_precision("f32")
    for (ga0, gb0, s0, w0), (ga1, gb1, s1, w1) in zip(res[False], res[True]):
        assert torch.equal(s0, s1)
        assert torch.equal(ga0, ga1), (ga0 - ga1).abs().max().item()      # the FIRST forward's backward saw its own activations
        assert torch.equal(gb0, gb1)
        for k in w0:
            assert torch.allclose(w0[k], w1[k], rtol=0, atol=2e-5 * float(w0[k].abs().max()) + 1e-12), k


def test_taped_training_steps_match_eager_steps():
    """Four full GAN steps (multi-stream, pipelined, dropout on) with the taped regions replaying from step 2 on, against the same
    steps run eagerly: same logged losses and the same weights afterwards, to the tolerance of two eager runs against each other
    (f32 atomics order + the discrete MAS path; see tests/test_gpu_graph.py)."""
    from optispeech_amd import tape
    la, sa, _ = _train(False)
    s0 = tape.stats()
    lb, sb, s1 = _train(True)
    assert s1["replayed"] - s0["replayed"] >= 3 * 16, (s0, s1)          # >= the 8 stacks' forward + backward in both phases, 3 steps
    for i, (x, y) in enumerate(zip(la, lb)):
        assert x.keys() == y.keys()
        for k in x:
            assert np.isfinite(y[k])
            assert abs(x[k] - y[k]) <= (6e-3 if i == 0 else 2e-2) * abs(x[k]) + 1e-4, (i, k, x[k], y[k])
    moved = 0
    for k in sa:
        if sa[k].is_floating_point():
            assert torch.allclose(sa[k], sb[k], rtol=1e-3, atol=1.5e-3), (k, (sa[k] - sb[k]).abs().max().item())
            moved += 1
    assert moved > 100


def test_taped_generator_segments_match_eager_steps():
    """The opt-in taped generator segments (OptiSpeech.tape_segments: acoustic model and vocoder as tape.Segment nodes -- forward AND
    backward replayed from call lists, dropout seed read from device memory) against the eager steps, same tolerance as above."""
    from optispeech_amd import tape
    la, sa, _ = _train(False)
    s0 = tape.stats()
    lb, sb, s1 = _train(True, segments=True)
    assert s1["recorded"] - s0["recorded"] >= 18, (s0, s1)              # the stacks' regions + 2 segments x (forward, backward)
    assert s1["poisoned"] == s0["poisoned"], "a generator segment fell back to eager execution"
    for i, (x, y) in enumerate(zip(la, lb)):
        for k in x:
            assert np.isfinite(y[k])
            assert abs(x[k] - y[k]) <= (6e-3 if i == 0 else 2e-2) * abs(x[k]) + 1e-4, (i, k, x[k], y[k])
    for k in sa:
        if sa[k].is_floating_point():
            assert torch.allclose(sa[k], sb[k], rtol=1e-3, atol=1.5e-3), (k, (sa[k] - sb[k]).abs().max().item())
