"""Flat parameter / gradient arena and the fused AdamW + global-norm clip step (reference row A18).

All parameters of one optimiser group live in ONE contiguous f32 buffer (and their gradients in another):
  * backward kernels accumulate straight into the gradient arena (``param.grad`` are views of it),
  * ``zero_grad`` is one memset, the gradient norm one reduction, the update one kernel,
  * data-parallel all-reduce works on arena slices (buckets) instead of ~250 small tensors.
"""
import math

import torch

from . import values

from ._lib import call


class FlatArena:
    def __init__(self, params, align=4):
        self.params = [p for p in params]
        assert self.params, "no parameters"
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
            offs.append(n)
            n += (p.numel() + align - 1) // align * align          # keep every tensor 16-byte aligned
        self.numel = n
        self.offsets = offs
        self.data = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        #: data-parallel reducer of this arena's gradients (set by the trainer); backward code that knows a parameter range is
        #: complete launches that range's all-reduce right away (optispeech_amd/dp.py: reduce_ready)
        self.reducer = None
        for p, o in zip(self.params, offs):
            self.data[o:o + p.numel()].copy_(p.detach().reshape(-1))
            p.data = self.data[o:o + p.numel()].view(p.shape)
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
            p._osp_arena = (self, o)

    def zero_grad(self):
        self.grad.zero_()

    def rebind_grads(self):
        """Re-attach ``param.grad`` views (after something set them to None)."""
        for p, o in zip(self.params, self.offsets):
            p.grad = self.grad[o:o + p.numel()].view(p.shape)


def cosine_warmup_factor(step, warmup, total, num_cycles=0.5):
    """transformers.get_cosine_schedule_with_warmup lambda (configs/model/scheduler/cosine_with_warmup.yaml)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    prog = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * prog)))


class CosineWarmupSchedule:
    """LambdaLR-like: ``lr = base_lr * factor(step)``; ``step()`` after every optimiser step (interval: step)."""

    def __init__(self, optimizer, num_warmup_steps=1000, num_training_steps=1_000_000, last_epoch=-1):
        self.opt, self.warmup, self.total = optimizer, num_warmup_steps, num_training_steps
        self.base_lr = optimizer.lr
        self.last_step = last_epoch + 1
        optimizer.lr = self.base_lr * cosine_warmup_factor(self.last_step, self.warmup, self.total)

    def step(self):
        self.last_step += 1
        self.opt.lr = self.base_lr * cosine_warmup_factor(self.last_step, self.warmup, self.total)

    def get_last_lr(self):
        return [self.opt.lr]


class FusedAdamW:
    """torch.optim.AdamW semantics (decoupled decay, bias correction) fused with clip_grad_norm_ over one arena."""

    def __init__(self, params, lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=1e-2):
        if isinstance(params, (list, tuple)) and params and isinstance(params[0], dict):
            params = [p for grp in params for p in grp["params"]]                 # reference passes param groups
        self.arena = params if isinstance(params, FlatArena) else FlatArena(list(params))
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.arena.data)
        self.exp_avg_sq = torch.zeros_like(self.arena.data)
        self.step_count = 0
        self._sumsq = torch.zeros(1, device=self.arena.data.device, dtype=torch.float64)
        #: (lr f32[1], step int64[1]) device tensors while the step is captured in / replayed from a hipGraph: the kernel
        #: then reads both from device memory (the replay loop refreshes them), not from its launch arguments
        self.dev_scalars = None

    def zero_grad(self):
        self.arena.zero_grad()

    def grad_sumsq(self):
        """device f64 scalar: sum of squares of the whole gradient arena."""
        self._sumsq.zero_()
        call("osp_sumsq", self.arena.grad, self.arena.numel, self._sumsq)
        return self._sumsq

    def step(self, max_norm=None, grad_scale=1.0):
        self.step_count += 1
        ss = self.grad_sumsq() if max_norm else None
        values.bump_param_epoch()                 # derived weight packs (disc_ops weight-norm cache) are stale from here on
        lr_dev, step_dev = self.dev_scalars if self.dev_scalars is not None else (None, None)
        call("osp_adamw_clip", self.arena.data, self.arena.grad, self.exp_avg, self.exp_avg_sq, self.arena.numel, ss,
             lr_dev, step_dev, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
             float(self.weight_decay), int(self.step_count), float(max_norm or 0.0), float(grad_scale))

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr = sd.get("lr", self.lr)
