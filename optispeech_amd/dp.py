"""Data parallelism for the training step: one process per GPU, utterance-batch sharding,
bucketed all-reduce of the flat gradient arenas over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm;
"gloo" in the CPU tests).

Overlap schedule (exactly equivalent to the reference's order of optimiser steps):
  G backward -> [G-grad all-reduce in flight] -> D forward/backward (needs neither G grads nor updated G weights)
  -> [D-grad all-reduce in flight] -> wait G -> clip + AdamW(G) -> wait D -> clip + AdamW(D).
Gradient averaging (1/world) is folded into the optimiser kernel's ``grad_scale``.
"""
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
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


class GradReducer:
    """Bucketed asynchronous all-reduce(sum) of a flat gradient buffer."""

    def __init__(self, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._pending = []
        # gloo on device tensors (the single-GPU test aid, OSP_DP_BACKEND=gloo) stages through the host from a worker thread
        # that synchronises streams on its own; with the step's side streams in flight that took seconds per collective on
        # the shared-GPU box (8-10 s/step vs 0.1 s), so the device is drained first.  RCCL orders on-stream: no host sync.
        self._drain_first = self.world > 1 and dist.get_backend(group) == "gloo"

    @property
    def active(self):
        return self.world > 1

    def start(self, flat_grad):
        """Launch the all-reduce of every bucket; returns immediately (work proceeds on RCCL's stream)."""
        if not self.active:
            return
        if self._drain_first and flat_grad.is_cuda:
            torch.cuda.synchronize()
        n = flat_grad.numel()
        for o in range(0, n, self.bucket_elems):
            w = dist.all_reduce(flat_grad[o:min(n, o + self.bucket_elems)], op=dist.ReduceOp.SUM, group=self.group,
                                async_op=True)
            self._pending.append(w)

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def broadcast_from_rank0(self, tensors):
        """Make every replica start from rank 0's values (parameter arenas, buffers)."""
        if not self.active:
            return
        if self._drain_first and any(t.is_cuda for t in tensors):
            torch.cuda.synchronize()
        for t in tensors:
            dist.broadcast(t, src=0, group=self.group)

    def mean_scalars(self, t):
        """In-place mean over ranks of a small packed tensor of log scalars (replaces ~20 sync_dist all-reduces)."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t /= self.world
        return t
