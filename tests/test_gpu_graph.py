"""hipGraph replay of the training step (BASELINE.json configs[1]): the captured step -- one graph on a
single GPU, the five-segment data-parallel stage order when forced -- must give what the eager step gives, including the
per-step scalars that live in device memory while replaying (dropout seed, AdamW step count, learning rate)."""
import numpy as np
import pytest
import torch

from tests._isolate import isolated

pytestmark = pytest.mark.gpu


def _run(mode, steps=3):
    from optispeech_amd import rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    cfg = ModelConfig()                                           # BASELINE widths, dropout / drop-path ON
    batch = synthetic_batch(2, 24, 96, cfg, seed=5, device="cuda")
    torch.manual_seed(3)
    torch.cuda.manual_seed(3)
    rng.manual_seed(3, 0)
    m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
    m.generator.segment_rand01 = torch.rand(2, generator=torch.Generator().manual_seed(1)).cuda()
    m.optimizers()
    for sch in m.lr_schedulers():                                 # short warm-up: the learning rate changes every step
        sch.warmup = 4
        sch.opt.lr = sch.base_lr * (1.0 / 4)
        sch.last_step = 1
    if mode == "hybrid":                                          # acoustic model + vocoder as graphed segments inside the eager step
        m.graph_segments = True
    elif mode != "eager":
        m.graph_steps = True
        m.graph_warmup_steps = 1
        m.graph_force_segments = mode == "segments"
    per_step = []
    for i in range(steps):
        m.training_step(batch, i)
        per_step.append(m.fetch_logs())
    torch.cuda.synchronize()
    opt_g, opt_d = m.optimizers()
    host = (m.global_step, opt_g.step_count, opt_d.step_count, opt_g.lr, opt_d.lr, rng.host_seed(),
            [s.last_step for s in m.lr_schedulers()])
    return per_step, {k: v.detach().clone() for k, v in m.state_dict().items()}, host, m


@pytest.mark.parametrize("mode", ["graph", "segments", "hybrid"])
@isolated
def test_graphed_step_matches_eager(mode):
    from optispeech_amd import precision
    precision.set_precision("bf16")
    try:
        la, sa, ha, _ = _run("eager")
        lb, sb, hb, m = _run(mode)
        if mode == "hybrid":
            assert len(m._gen_segments) == 1 and not m._step_graphs
        else:
            assert len(m._step_graphs) == 1
            sg = next(iter(m._step_graphs.values()))
            assert len(sg.graphs) == (1 if mode == "graph" else 5)
        assert ha == hb, (ha, hb)                                 # step counts, schedule position, lr, seed: same bookkeeping
        # every step of the graph run is a replay (the capture's warm-up is rolled back)
        for i, (x, y) in enumerate(zip(la, lb)):
            assert x.keys() == y.keys()
            for k in x:
                # tolerance of two eager runs against each other: f32 atomics order + the discrete MAS path amplify round-off
                # (step 0 observed up to 2.7e-3 on the pitch loss between an eager and a replayed run of identical inputs)
                assert abs(x[k] - y[k]) <= (6e-3 if i == 0 else 2e-2) * abs(x[k]) + 1e-4, (i, k, x[k], y[k])
        moved = 0
        for k in sa:
            if sa[k].is_floating_point():
                assert torch.allclose(sa[k], sb[k], rtol=1e-3, atol=1.5e-3), (k, (sa[k] - sb[k]).abs().max().item())
                moved += 1
        assert moved > 100
    finally:
        precision.set_precision("f32")


@isolated
def test_graph_replay_advances_dropout_seed_and_weights():
    """Replays are not re-runs of the captured step: the dropout masks change (seed in device memory), the weights keep moving
    and the logged losses change from replay to replay."""
    from optispeech_amd import precision
    precision.set_precision("bf16")
    try:
        logs, _, _, m = _run("graph", steps=4)
        vals = [l["total_loss/generator"] for l in logs]
        assert all(np.isfinite(v) for v in vals)
        assert len({round(v, 6) for v in vals[1:]}) == 3, vals     # three replays, three different losses
    finally:
        precision.set_precision("f32")


@isolated
def test_graph_captured_decode_equals_eager_synthesise(golden):
    """BASELINE.json configs[4]: synthesise() with the decode replayed from hipGraphs returns exactly the eager result, for
    two different batches through the same and through a new capture."""
    from oracle import schema as S
    from optispeech_amd import precision
    from optispeech_amd.config import make_generator
    from tests.test_gpu_generator import _small_cfg
    precision.set_precision("bf16")
    try:
        g = golden("synth_small")
        gen = make_generator(_small_cfg()).to("cuda").eval()
        W = S.make_weights(S.generator_schema(S.SMALL), int(g["seed"]))
        W["generator.duration_predictor.linear.bias"].fill_(float(g["dur_bias"]))
        gen.load_state_dict({k[len("generator."):]: v for k, v in W.items()})
        x, xl = torch.from_numpy(g["in_x"]).cuda(), torch.from_numpy(g["in_x_lengths"])
        x2 = x.flip(0).contiguous()
        xl2 = xl.flip(0).contiguous()
        outs = {}
        for graph in (False, True, True):
            gen.graph_decode = graph
            a = gen.synthesise(x, xl, d_factor=1.1, p_factor=1.6, e_factor=1.2)
            b = gen.synthesise(x2, xl2, d_factor=1.3, p_factor=1.0, e_factor=1.0)
            outs.setdefault(graph, []).append((a, b))
        assert len(gen._decode_graphs) >= 1
        (ea, eb) = outs[False][0]
        for ga, gb in outs[True]:
            assert np.array_equal(ga["durations"].numpy(), ea["durations"].numpy()) and np.array_equal(ga["durations"].numpy(), g["durations"])
            assert torch.equal(ga["wav"], ea["wav"]) and torch.equal(gb["wav"], eb["wav"])
            assert ga["rtf"] > 0 and ga["latency"] > 0
    finally:
        precision.set_precision("f32")
