"""Data parallelism for the training step: one process per GPU, utterance-batch sharding,
bucketed all-reduce of the flat gradient arenas over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm;
"gloo" in the CPU tests).

Overlap schedule (exactly equivalent to the reference's order of optimiser steps):
  G backward -> [G-grad all-reduce in flight] -> D forward/backward (needs neither G grads nor updated G weights)
  -> [D-grad all-reduce in flight] -> wait G -> clip + AdamW(G) -> wait D -> clip + AdamW(D).
Gradient averaging (1/world) is folded into the optimiser kernel's ``grad_scale``.
"""
import contextlib
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun contract). Returns (world, rank, local)."""
    world, rank, local = env_world()
    if os.environ.get("OSP_DP_SINGLE_DEVICE") == "1":        # test aid: all ranks on GPU 0 (gloo moves the buckets via the host)
        local = 0
    if torch.cuda.is_available():
        torch.cuda.set_device(local)                # every backend: the kernels launch on the CURRENT device's stream
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("OSP_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        native = backend == "native"                  # gradients over the C-ABI RCCL communicator, gloo as the control plane
        dist.init_process_group(backend="gloo" if native else backend, rank=rank, world_size=world)
        if native:
            init_native_comm()
    return world, rank, local


# ---------------------------------------------------------------------------------------------- native communicator (C ABI)
_native = {"ready": False, "stream": None}


def init_native_comm(group=None):
    """Create the RCCL communicator behind the C ABI (osp_comm_init, csrc/comm.cpp): rank 0 draws the unique id, the existing
    torch.distributed group (any backend; gloo is enough) only carries those 128 bytes.  Gradient traffic then bypasses
    torch.distributed altogether: osp_allreduce_bucket on a dedicated HIP stream."""
    import ctypes
    from ._lib import OspError, lib
    cd = lib().cdll
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    buf = (ctypes.c_char * 128)()
    if rank == 0 and cd.osp_comm_unique_id(buf) != 0:
        raise OspError("osp_comm_unique_id: " + cd.osp_last_error().decode())
    if world > 1:
        t = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0, group=group)
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(t.tolist()))
    if cd.osp_comm_init(ctypes.c_int64(rank), ctypes.c_int64(world), buf) != 0:
        raise OspError("osp_comm_init: " + cd.osp_last_error().decode())
    _native["ready"], _native["stream"] = True, torch.cuda.Stream()
    return world


def destroy_native_comm():
    from ._lib import lib
    if _native["ready"]:
        torch.cuda.synchronize()
        lib().cdll.osp_comm_destroy()
        _native["ready"], _native["stream"] = False, None


#: Context-manager factory entered around every BLOCKING host-side collective of the gloo path (the single-GPU test aid stages
#: device buffers through the host).  Nothing in a real run (one process per GPU, RCCL ordered on-stream); tests/test_gpu_dp.py, whose
#: two ranks share one GPU, installs a turn-taking lock here.
around_host_collective = contextlib.nullcontext


class _NativeWork:
    """Handle of one osp_allreduce_bucket launch: ``wait()`` orders the CURRENT stream behind it (no host sync)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class GradReducer:
    """Bucketed asynchronous all-reduce(sum) of a flat gradient buffer."""

    def __init__(self, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        #: gradient buckets go through the C-ABI communicator (RCCL, osp_allreduce_bucket) instead of torch.distributed
        self.native = _native["ready"]
        if self.native:
            from ._lib import lib
            self.world = int(lib().cdll.osp_comm_world())
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._pending = []
        self._covered = []
        # gloo on device tensors (the single-GPU test aid, OSP_DP_BACKEND=gloo) stages through the host from a worker thread
        # that synchronises streams on its own; with the step's side streams in flight that took seconds per collective on
        # the shared-GPU box (8-10 s/step vs 0.1 s), so the device is drained first.  RCCL orders on-stream: no host sync.
        gloo = self.world > 1 and dist.is_initialized() and dist.get_backend(group) == "gloo"
        self._drain_first = gloo and not self.native
        self._ctrl_drain = gloo                      # small control-plane collectives (broadcast, log scalars) on device tensors
        self._force_active = False
        #: measurement aid (bench.py): bracket every wait() with two events on the waiting stream; exposed_ms() = how long that
        #: stream stood still for the collectives (0 when they finished under the compute they overlap)
        self.measure = False
        self._brackets = []

    def exposed_ms(self):
        """Sum over the bracketed wait() calls of the time the waiting stream stalled (call after a device synchronise)."""
        tot = sum(a.elapsed_time(b) for a, b in self._brackets)
        self._brackets.clear()
        return tot

    @property
    def active(self):
        return self.world > 1 or self._force_active        # (_force_active: single-rank exercise of the native path in tests)

    def start(self, flat_grad):
        """Launch the all-reduce of every bucket; returns immediately (work proceeds on RCCL's stream)."""
        if not self.active:
            return
        if self._drain_first and flat_grad.is_cuda:
            # gloo on device tensors = the single-GPU test aid (OSP_DP_BACKEND=gloo).  The buffer is staged through the host HERE,
            # synchronously: gloo's own device path (private streams, worker threads) produced rare wrong slices (1e-3 of a
            # sub-discriminator's gradients, replicas still identical) when the two ranks shared one GPU, while the same step is
            # reproducible to 1e-7 run to run in one process (tools/determinism_probe.py (git history), also under contention).
            torch.cuda.synchronize()
            host = flat_grad.detach().to("cpu")
            with around_host_collective():
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.copy_(host)
            torch.cuda.synchronize()
            return
        n = flat_grad.numel()
        if flat_grad.is_cuda:
            # weight-gradient kernels of a running backward pass may still be queued on ops.side_wgrad's streams, which the
            # calling stream only joins at the end of that pass: a gradient-ready collective issued from inside it waits here
            from .ops import wgrad_side_streams
            for s in wgrad_side_streams():
                torch.cuda.current_stream().wait_stream(s)
        if self.native:
            from ._lib import call
            cs = _native["stream"]
            cs.wait_stream(torch.cuda.current_stream())          # the bucket's producers are on the calling stream
            with torch.cuda.stream(cs):
                for o in range(0, n, self.bucket_elems):
                    call("osp_allreduce_bucket", flat_grad[o:min(n, o + self.bucket_elems)], min(n, o + self.bucket_elems) - o)
                self._pending.append(_NativeWork(cs.record_event()))
            flat_grad.record_stream(cs)
            return
        for o in range(0, n, self.bucket_elems):
            w = dist.all_reduce(flat_grad[o:min(n, o + self.bucket_elems)], op=dist.ReduceOp.SUM, group=self.group,
                                async_op=True)
            self._pending.append(w)

    def start_range(self, flat_grad, lo, hi):
        """All-reduce ``flat_grad[lo:hi]`` now (its gradients are complete) and remember the interval, so that the closing
        ``start_rest`` only launches what is left.  Called from inside the backward pass, on whatever stream produced the range:
        the collective is ordered behind that stream's work and overlaps the rest of the backward."""
        if not self.active or hi <= lo:
            return
        self._covered.append((int(lo), int(hi)))
        self.start(flat_grad[lo:hi])

    def start_rest(self, flat_grad):
        """Launch the all-reduce of every element not yet covered by start_range since the last wait()."""
        if not self.active:
            return
        pos, n = 0, flat_grad.numel()
        for lo, hi in sorted(self._covered):
            if lo > pos:
                self.start(flat_grad[pos:lo])
            pos = max(pos, hi)
        if pos < n:
            self.start(flat_grad[pos:n])
        self._covered.clear()

    def wait(self):
        bracket = self.measure and self._pending and torch.cuda.is_available()
        if bracket:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._pending:
            w.wait()
        if bracket:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._brackets.append((e0, e1))
        self._pending.clear()
        self._covered.clear()

    def broadcast_from_rank0(self, tensors):
        """Make every replica start from rank 0's values (parameter arenas, buffers)."""
        if not self.active:
            return
        if self._ctrl_drain and any(t.is_cuda for t in tensors):
            # gloo (test aid): through the host, so that the collective itself involves no device work of either rank
            torch.cuda.synchronize()
            for t in tensors:
                host = t.detach().to("cpu")
                with around_host_collective():
                    dist.broadcast(host, src=0, group=self.group)
                t.copy_(host)
            torch.cuda.synchronize()
            return
        for t in tensors:
            dist.broadcast(t, src=0, group=self.group)

    def mean_scalars(self, t):
        """In-place mean over ranks of a small packed tensor of log scalars (replaces ~20 sync_dist all-reduces)."""
        if self.world > 1:
            if self._ctrl_drain and t.is_cuda:
                torch.cuda.synchronize()
                host = t.detach().to("cpu")
                with around_host_collective():
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(host)
                t /= self.world
                return t
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t /= self.world
        return t


def reduce_ready(params):
    """Gradient-ready hook for code that writes parameter gradients straight into a flat arena (no AccumulateGrad node to hang a
    DDP hook on): ``params`` are complete, so if they live in an arena with an active reducer, all-reduce their slice now.  The
    arena slots of one sub-module are contiguous (registration order), so the slice is [min offset, max end)."""
    lo = hi = None
    arena = None
    for p in params:
        ar = getattr(p, "_osp_arena", None)
        if ar is None or not p.requires_grad:
            return
        if arena is None:
            arena = ar[0]
        elif arena is not ar[0]:
            return
        lo = ar[1] if lo is None else min(lo, ar[1])
        end = ar[1] + (p.numel() + 3) // 4 * 4
        hi = end if hi is None else max(hi, end)
    red = getattr(arena, "reducer", None) if arena is not None else None
    if red is None or not red.active or not getattr(red, "eager_ranges", True):
        return
    red.start_range(arena.grad, lo, min(hi, arena.numel))
