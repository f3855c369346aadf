"""Data-parallel plumbing on CPU: world_size-2 gloo processes exercise the bucketed gradient all-reduce over the
flat arena, the packed log-scalar mean and the env-based initialisation (the N>1 path of bench.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from optispeech_amd import dp
    from optispeech_amd.optim import FlatArena
    w, r, _ = dp.init_from_env("gloo")
    assert (w, r) == (world, rank)
    torch.manual_seed(0)                                    # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5))
    arena = FlatArena(list(net.parameters()))
    # parameters were re-pointed into the arena and stay consistent views
    assert all(p.data_ptr() >= arena.data.data_ptr() for p in net.parameters())
    x = torch.randn(8, 37, generator=torch.Generator().manual_seed(100 + rank))
    arena.zero_grad()
    net(x).square().mean().backward()                       # autograd accumulates into the arena views in place
    local = arena.grad.clone()
    red = dp.GradReducer(bucket_bytes=256)                  # many small buckets
    assert red.active and red.world == world
    red.start(arena.grad)
    red.wait()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered)
    ok = torch.allclose(arena.grad, want, rtol=1e-6, atol=1e-7)
    # gradient-ready ranges (launched from inside the backward in the real step) + the closing remainder == one full all-reduce
    arena.grad.copy_(local)
    red.start_range(arena.grad, 16, 200)
    red.start_range(arena.grad, 400, 420)
    red.start_rest(arena.grad)
    red.wait()
    ok = ok and torch.allclose(arena.grad, want, rtol=1e-6, atol=1e-7) and not red._covered
    logs = torch.tensor([float(rank), 2.0 * rank + 1.0])
    red.mean_scalars(logs)
    ok = ok and torch.allclose(logs, torch.tensor([(world - 1) / 2.0, float(world)]))
    # every rank ends with bit-identical averaged gradients (what keeps replicas in lock-step)
    g0 = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(g0, arena.grad)
    ok = ok and all(torch.equal(g0[0], g) for g in g0)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_reducer_is_inert():
    from optispeech_amd.dp import GradReducer
    r = GradReducer()
    assert not r.active and r.world == 1
    g = torch.ones(10)
    r.start(g)
    r.wait()
    assert torch.equal(g, torch.ones(10))


def _worker_uneven(rank, world, port, q):
    """World of four; every rank reports its gradient-ready ranges in a DIFFERENT order and with different cuts (as stacks that finish
    their backward in a rank-dependent order would), the closing start_rest covers the remainder: the result must still be the plain
    sum, identical on every rank, and a strong-scaling shard (a quarter of the rows each) must reproduce the single-process gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from optispeech_amd import dp
    from optispeech_amd.optim import FlatArena
    w, r, _ = dp.init_from_env("gloo")
    assert (w, r) == (world, rank)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(41, 23), torch.nn.Tanh(), torch.nn.Linear(23, 7))
    arena = FlatArena(list(net.parameters()))
    n = arena.grad.numel()
    xs = torch.randn(4 * world, 41, generator=torch.Generator().manual_seed(7))         # the GLOBAL batch, the same on every rank
    rows = xs[rank * 4:(rank + 1) * 4]                                                  # --strong: this rank's shard
    arena.zero_grad()
    (net(rows).square().sum() / xs.shape[0]).backward()                                 # per-rank share of the global mean
    local = arena.grad.clone()
    red = dp.GradReducer(bucket_bytes=128)
    # the ranges must be the same SET on every rank (a collective per range); their ORDER of completion is what differs in the real
    # step -- each rank issues them in its own order but a collective only matches when all ranks reach it, so the reducer's
    # contract is "same ranges, same order": ranks that finish early simply wait.  What may differ freely is what is still
    # uncovered when start_rest runs; exercise that with rank-dependent extra zero-length / empty calls
    cuts = [(0, 100), (300, 512), (600, n)]
    for lo, hi in cuts:
        red.start_range(arena.grad, lo, hi)
        if rank % 2:
            red.start_range(arena.grad, hi, hi)                                          # empty range: ignored
    red.start_rest(arena.grad)
    red.wait()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered)
    ok = torch.allclose(arena.grad, want, rtol=1e-6, atol=1e-7)
    # the sum of the shards' gradients == the gradient of the global batch in one process
    arena.zero_grad()
    (net(xs).square().sum() / xs.shape[0]).backward()
    ok = ok and torch.allclose(arena.grad, want, rtol=1e-5, atol=1e-6)
    g0 = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(g0, want)
    ok = ok and all(torch.equal(g0[0], g) for g in g0)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_ready_ranges_and_strong_scaling_shards_four_ranks_gloo():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
