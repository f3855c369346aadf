"""Data parallelism on the REAL training step (the reference's DDP contract configs/trainer/ddp.yaml:4-9,
reducers at base_lightning_module.py:99,119): two ranks (both on GPU 0, gloo moving the buckets through the host -- RCCL needs
one GPU per rank) run ``OptiSpeech.training_step`` on two different micro-batches.  8e's own parity target:

    the N-rank result equals the mean of the N single-rank gradient sets, and replicas stay identical after the update.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests._isolate import isolated

pytestmark = pytest.mark.gpu

#: |all-reduced sum / world - mean of the single-rank gradient sets| / |mean|, per arena.  Same kernels, same inputs; the only
#: run-to-run freedom is the order of f32 atomics in the split-K weight gradients.
TOL = 2e-5


import contextlib

_TURN = [None]


class _yield_turn:
    """``with _yield_turn():`` around a blocking gloo collective: drain the device, let the other rank compute, take the GPU back.
    Test aid (two PROCESSES computing on one MI355X of this pool at the same time give occasionally different FFT results):
    installed as ``dp.around_host_collective`` by the two-rank worker below."""

    def __enter__(self):
        import torch
        self.lock = _TURN[0]
        if self.lock is not None:
            torch.cuda.synchronize()
            self.lock.release()
        return self

    def __exit__(self, *exc):
        if self.lock is not None:
            self.lock.acquire()
        return False


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed, rank_seed_offset=0):
    from oracle import schema as S
    from optispeech_amd import rng
    from optispeech_amd.config import ModelConfig, make_optispeech
    c = S.SMALL
    cfg = ModelConfig(dim=c.dim, enc_inter=c.enc_inter, dec_inter=c.dec_inter, dur=c.dur + (0.0,), pitch=c.pitch + (0.0,),
                      energy=c.energy + (0.0,), voc_dim=c.voc_dim, voc_inter=c.voc_inter, voc_layers=c.voc_layers).no_dropout()
    torch.manual_seed(seed + rank_seed_offset)              # rank_seed_offset != 0: replicas that did NOT seed identically
    rng.manual_seed(seed, 0)
    m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
    return cfg, m


def _batches(cfg):
    from optispeech_amd.config import synthetic_batch
    out = []
    for r in range(2):
        b = synthetic_batch(2, 24, 96, cfg, seed=50 + r, ragged=True, device="cuda")
        out.append((b, torch.tensor([0.25 + 0.5 * r, 0.6 - 0.3 * r], device="cuda")))
    return out


def _grads_of_one_step(m, batch, r01):
    """(G-arena gradient, D-arena gradient) of one training step from the CURRENT weights, without updating them."""
    got = {}
    og, od = m.optimizers()
    for name, o in (("g", og), ("d", od)):
        o.step = (lambda n, oo: (lambda *a, **k: got.__setitem__(n, oo.arena.grad.detach().clone())))(name, o)
    m.generator.segment_rand01 = r01
    m.training_step(batch, 0)
    torch.cuda.synchronize()
    return got["g"], got["d"]


def _worker(rank, world, port, graph, q, lock):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), OSP_DP_SINGLE_DEVICE="1", OSP_DP_BACKEND="gloo")
        import torch.distributed as dist
        from optispeech_amd import dp, precision
        precision.set_precision("f32")
        # ---- single-rank references, before the process group exists (their reducers are inert)
        with lock:                                        # one rank at a time on the shared GPU (see the test's docstring)
            cfg, ref = _build(7)
            batches = _batches(cfg)
            ref.optimizers()
            assert not ref._reducers[0].active
            w0 = [o.arena.data.clone() for o in ref.optimizers()]
            singles = [_grads_of_one_step(ref, b, r01) for b, r01 in batches]
            mean_g = (singles[0][0] + singles[1][0]) / 2
            mean_d = (singles[0][1] + singles[1][1]) / 2
            torch.cuda.synchronize()
        # ---- the two-rank run: rank r sees micro-batch r; rank 1 deliberately starts from DIFFERENT weights, which the
        # broadcast from rank 0 in optimizers() must repair
        w, r, _ = dp.init_from_env()
        assert (w, r) == (world, rank) and dist.get_backend() == "gloo"
        # from here on the ranks take TURNS on the one GPU they share: a rank computes while it holds the lock and hands it over,
        # device drained, whenever it blocks in a gloo collective (_yield_turn below, installed as dp.around_host_collective; see the test's docstring)
        lock.acquire()
        _TURN[0] = lock
        dp.around_host_collective = _yield_turn
        cfg, m = _build(7, rank_seed_offset=rank)
        m.graph_steps = graph
        m.graph_warmup_steps = 1
        og, od = m.optimizers()
        for sch in m.lr_schedulers():                     # no warm-up (its first step has lr 0): the update must move the weights
            sch.warmup = 0
            sch.opt.lr = sch.base_lr
        assert m._reducers[0].active and m._reducers[0].world == 2
        assert torch.equal(og.arena.data, w0[0]) and torch.equal(od.arena.data, w0[1]), "replicas do not start from rank 0's weights"
        b, r01 = batches[rank]
        m.generator.segment_rand01 = r01
        # this rank's own (pre-reduce) gradients, captured where the reducer takes them: they must equal the single-rank reference
        local = {}
        if not graph:
            for name, red, o in (("g", m._reducers[0], og), ("d", m._reducers[1], od)):
                def hook(flat, _n=name, _orig=red.start_rest):
                    torch.cuda.synchronize()
                    local[_n] = flat.detach().clone()
                    return _orig(flat)
                red.start_rest = hook
        m.training_step(b, 0)
        logs = m.fetch_logs()
        torch.cuda.synchronize()
        ok = True
        msgs = []
        for i_, name in enumerate(("g", "d")):
            if name in local:
                e = ((local[name] - singles[rank][i_]).norm() / singles[rank][i_].norm()).item()
                msgs.append(f"{name}: this rank's pre-reduce gradients vs its single-rank reference = {e:.2e}")
        # after the step the gradient arenas still hold what the update consumed: the all-reduced SUM (1/world is folded into
        # the update kernel's grad_scale)
        for name, o, mean in (("g", og, mean_g), ("d", od, mean_d)):
            got = o.arena.grad.detach() / world
            err = ((got - mean).norm() / mean.norm()).item()
            msgs.append(f"{name}: |sum/world - mean of single-rank grads| / |mean| = {err:.2e}")
            if err >= 6e-4:                               # where: one contiguous slice (a collective / range problem) or scattered?
                bad = ((got - mean).abs() > 1e-4 * mean.abs().max()).nonzero().flatten()
                msgs.append(f"{name}: {bad.numel()} of {got.numel()} elements off, index range [{int(bad.min()) if bad.numel() else -1}, "
                            f"{int(bad.max()) if bad.numel() else -1}], got/mean norm ratio {(got.norm() / mean.norm()).item():.6f}")
                by = {id(p): n for n, p in m.named_parameters()}
                worst = sorted(((((got - mean)[off:off + p.numel()]).norm().item(), by[id(p)]) for p, off in zip(o.arena.params, o.arena.offsets)),
                               reverse=True)[:4]
                msgs.append(f"{name}: largest deviations " + ", ".join(f"{n} {v:.2e}" for v, n in worst))
            # two runs of the SAME step differ only by the f32-atomic order of the split-K weight gradients; a missing / doubled /
            # unscaled contribution would show as O(1)
            ok = ok and err < TOL
        # replicas identical after the update: compare every rank's arenas bit for bit
        for o in (og, od):
            mine = o.arena.data.detach().cpu()
            both = [torch.empty_like(mine) for _ in range(world)]
            with _yield_turn():
                dist.all_gather(both, mine)
            mine = mine.to("cuda")
            same = torch.equal(both[0], both[1])
            ok = ok and same
            msgs.append(f"replicas bit-identical: {same}")
            moved = (mine - w0[0 if o is og else 1]).abs().max().item()
            ok = ok and moved > 0
        q.put((rank, bool(ok), msgs, {k: float(v) for k, v in logs.items() if k.startswith("total_loss")}))
        with _yield_turn():
            dist.barrier()
        dp.around_host_collective = contextlib.nullcontext
        _TURN[0] = None
        lock.release()
        dist.destroy_process_group()
    except Exception as e:                                # noqa: BLE001
        import traceback
        q.put((rank, False, [traceback.format_exc()], {}))
        if _TURN[0] is not None:                          # do not leave the other rank waiting for its turn
            _TURN[0] = None
            try:
                lock.release()
            except ValueError:
                pass
        raise


def _attempt(graph):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q, lock = ctx.Queue(), ctx.Lock()
    procs = [ctx.Process(target=_worker, args=(r, world, port, graph, q, lock)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    res.sort()
    return res, [p.exitcode for p in procs]


@pytest.mark.parametrize("graph", [False, True])
def test_two_rank_training_step_equals_mean_of_single_rank_gradients(graph):
    """ONE attempt, strict.  Both ranks share ONE GPU here (RCCL needs a GPU per rank, the box has one).  Two PROCESSES computing
    on one MI355X of this pool at the same time is not a configuration the product runs in, and it makes FFT results come out
    different in a few percent of the repetitions -- rocFFT behind torch.stft as well as this package's STFT kernel, also in a
    stand-alone HIP program with no torch in it, never in a process that has the GPU to itself (tools/probes/shared_gpu_all.sh (git history),
    profiles/r03_shared_gpu_probe.txt).  So the two ranks take TURNS on the GPU (_yield_turn: a rank
    hands the GPU over, drained, whenever it blocks in a gloo collective): every kernel of either rank then runs with the GPU to
    itself, and the comparison is strict again -- a single attempt, no retry."""
    res, codes = _attempt(graph)
    bad = [(rank, msgs) for rank, ok, msgs, logs in res if not ok]
    if bad:
        pytest.fail("\n".join(f"rank {rank}:\n  " + "\n  ".join(str(x) for x in msgs) for rank, msgs in bad), pytrace=False)
    for rank, ok, msgs, logs in res:
        print(f"rank {rank}: " + "; ".join(str(x) for x in msgs))
    # the logged losses are the mean over ranks (one packed all-reduce): identical on both
    assert res[0][3] == res[1][3] and all(np.isfinite(v) for v in res[0][3].values())
    assert codes == [0, 0]


@isolated
def test_native_comm_c_abi_single_rank(tmp_path):
    """The RCCL communicator behind the C ABI (csrc/comm.cpp) on this box's one GPU: (1) the C++ host example of
    tests/native/comm_example.cpp runs (unique id -> init -> all-reduce -> destroy, sum over 1 rank = identity), (2) the same entry
    points through GradReducer's native path (OSP_DP_BACKEND=native), with gradient-ready ranges + remainder."""
    import subprocess
    from tests.test_abi import _build_comm_example
    exe = _build_comm_example(os.path.join(tmp_path, "comm_example"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.returncode, r.stdout, r.stderr)
    from optispeech_amd import dp
    world = dp.init_native_comm()
    try:
        assert world == 1
        red = dp.GradReducer(bucket_bytes=1 << 20)
        assert red.native and red.world == 1 and not red.active
        red._force_active = True
        g = torch.randn(3_000_001, device="cuda")
        want = g.clone()
        red.start_range(g, 1024, 500_000)
        red.start_rest(g)
        red.wait()
        torch.cuda.synchronize()
        assert torch.equal(g, want)
    finally:
        dp.destroy_native_comm()
