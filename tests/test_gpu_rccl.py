"""RCCL on the one GPU a test box has (the reference's DDP contract configs/trainer/ddp.yaml:4-9, reducers at
base_lightning_module.py:99,119): the REAL ``training_step`` -- eight sub-discriminator streams, the vocoder stream, the side
weight-gradient streams, gradient-ready range collectives issued from inside the backward, ``pipeline_steps`` -- with the
gradient reducers FORCED ACTIVE on a world of one rank, through both transports:

  * ``torch.distributed`` backend "nccl" (= RCCL): bucketed async all-reduce on RCCL's stream, ``work.wait()`` on the caller's;
  * the C-ABI communicator (``osp_comm_*`` / ``osp_allreduce_bucket``, csrc/comm.cpp) on its dedicated stream.

A sum over one rank is the identity and grad_scale = 1 / world = 1, so three such steps must reproduce three steps of the same
model without data parallelism: same logged losses, same arenas (to the run-to-run freedom of the step itself: f32 atomics in
the split-K weight gradients), identical host bookkeeping.  What this exercises is the ON-STREAM ORDERING of the collectives
against the step's streams, which nothing else runs on a 1-GPU box.  Each case runs in a child interpreter (RCCL creates
proxy threads and its own streams; a runtime fault there must be a named failure, tests/_isolate.py).
"""
import os
import socket

import numpy as np
import pytest
import torch

from tests._isolate import isolated

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed=11):
    from optispeech_amd import precision, rng
    precision.set_precision("bf16")       # the production schedule: sub-discriminator streams + gradient-ready ranges exist in this mode
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    cfg = ModelConfig()                                          # BASELINE widths; dropout / drop-path on
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    rng.manual_seed(seed, 0)
    m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
    batch = synthetic_batch(2, 24, 96, cfg, seed=5, device="cuda")
    m.generator.segment_rand01 = torch.tensor([0.3, 0.6], device="cuda")
    for sch in m.lr_schedulers():
        sch.warmup = 0
        sch.opt.lr = sch.base_lr
    return m, batch


def _three_steps(m, batch, pipeline):
    m.pipeline_steps = pipeline
    logs = []
    for i in range(3):
        m.training_step(batch, i)
        logs.append(m.fetch_logs())
    m.join()
    torch.cuda.synchronize()
    og, od = m.optimizers()
    return logs, og.arena.data.clone(), od.arena.data.clone(), (og.step_count, od.step_count, m.global_step)


def _compare(a, b):
    (la, ga, da, ha), (lb, gb, db, hb) = a, b
    assert ha == hb, (ha, hb)
    for x, y in zip(la, lb):
        assert x.keys() == y.keys()
        for k in x:
            assert np.isfinite(x[k]) and abs(x[k] - y[k]) <= 2e-3 * max(1.0, abs(y[k])), (k, x[k], y[k])
    for name, u, v in (("generator", ga, gb), ("discriminator", da, db)):
        err = ((u - v).norm() / v.norm()).item()
        print(f"{name}: arena deviation after three steps {err:.2e}")
        # two runs of the SAME three steps differ by 1e-4 .. 2.5e-4 here (measured): f32 atomic order in the split-K weight gradients,
        # which Adam's first steps turn into +-lr on near-zero gradients.  What this test is for -- a hang, a fault, a collective
        # ordered before its producers or after its consumer -- does not hide in that: it shows as NaN / O(1e-2..1) / no return.
        assert err < 1e-3, (name, err)


@pytest.mark.parametrize("backend", ["nccl", "native"])
@pytest.mark.parametrize("pipeline", [False, True])
@isolated
def test_real_step_with_forced_reducers_on_one_rank(backend, pipeline):
    import torch.distributed as dist
    from optispeech_amd import dp
    ref = _three_steps(*_model(), pipeline)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl"
    else:
        assert dp.init_native_comm() == 1
    try:
        m, batch = _model()
        og, od = m.optimizers()
        calls = {"start": 0, "range": 0}
        for red in m._reducers:
            assert red.world == 1 and red.native == (backend == "native") and not red.active
            red._force_active = True
            assert red.active
            start, start_range = red.start, red.start_range

            def counted_start(flat, _s=start):
                calls["start"] += 1
                return _s(flat)

            def counted_range(flat, lo, hi, _s=start_range):
                calls["range"] += 1
                return _s(flat, lo, hi)
            red.start, red.start_range = counted_start, counted_range
        got = _three_steps(m, batch, pipeline)
        # collectives really ran: per step the 8 sub-discriminator slices from inside the backward + the remainders
        assert calls["range"] >= 3 * 8 and calls["start"] > calls["range"], calls
        _compare(got, ref)
    finally:
        torch.cuda.synchronize()
        if backend == "nccl":
            dist.destroy_process_group()
        else:
            dp.destroy_native_comm()


@isolated
def test_bench_single_rank_under_torchrun(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the launcher path of the driver's scaling run
    (env rendezvous on 127.0.0.1, LOCAL_RANK device selection, rank-0 JSON line) on the one GPU available."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-infer", "--no-am-only"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["roofline"]["frac"] > 0


@isolated
def test_graphed_pretraining_step_waits_for_the_generator_all_reduce():
    """under data parallelism with ``graph_steps`` in the pre-training regime (global_step < pretraining_steps)
    the generator all-reduce was started after the [G forward + backward] graph and never waited for before the AdamW(G) graph.
    Here: the segmented (data-parallel) graph order on one rank with the reducers forced active over the C-ABI communicator --
    after every step nothing may be left pending, and the replayed steps must match the eager pre-training steps."""
    from optispeech_amd import dp
    assert dp.init_native_comm() == 1
    try:
        res = {}
        for graph in (False, True):
            m, batch = _model()
            m.train_args.pretraining_steps = 1 << 60
            m.optimizers()
            for red in m._reducers:
                red._force_active = True
            waits = {"n": 0}
            rg = m._reducers[0]
            w0 = rg.wait

            def counted_wait(_w=w0):
                waits["n"] += 1
                return _w()
            rg.wait = counted_wait
            m.graph_steps, m.graph_warmup_steps = graph, 1
            logs = []
            for i in range(3):
                m.training_step(batch, i)
                assert not rg._pending, "generator all-reduce still pending after the step"
                logs.append(m.fetch_logs())
            torch.cuda.synchronize()
            assert waits["n"] >= 3, waits
            og, od = m.optimizers()
            assert od.step_count == 0 and og.step_count == 3                     # no discriminator phase in this regime
            res[graph] = (logs, og.arena.data.clone())
        for a, b in zip(res[False][0], res[True][0]):
            assert a.keys() == b.keys() and all(np.isfinite(v) for v in b.values())
            for k in a:
                assert abs(a[k] - b[k]) <= 2e-3 * max(1.0, abs(a[k])), (k, a[k], b[k])
        err = ((res[True][1] - res[False][1]).norm() / res[False][1].norm()).item()
        assert err < 1e-3, err
    finally:
        torch.cuda.synchronize()
        dp.destroy_native_comm()
