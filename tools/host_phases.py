#!/usr/bin/env python3
"""Host (enqueue) time of the training step by phase, no device synchronisation inside the step (diagnostic)."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision, _lib, optim
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
dev = "cuda"
torch.manual_seed(0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to(dev).train()
m.pipeline_steps = True
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device=dev)
m.optimizers()
acc = collections.defaultdict(float)
cnt = collections.Counter()


def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[key] += time.perf_counter() - t
    setattr(obj, name, g)


wrap(m, "training_step_g", "G forward (generator + disc on gen)")
wrap(m, "training_step_d", "D forward")
wrap(m.discriminator, "prepare_disc_inputs", "prepare_disc_inputs")
wrap(torch.Tensor, "backward", "backward (G + D)")
wrap(optim.FusedAdamW, "step", "optimizer steps")
wrap(optim.FusedAdamW, "zero_grad", "zero_grad")
wrap(m, "_process_batch", "  of G forward: generator")
wrap(m.discriminator, "forward_gen", "  of G forward: forward_gen")
lib = _lib.lib(); orig = lib.call
def call(name, *a):
    t = time.perf_counter(); orig(name, *a); acc["C-ABI calls (all threads)"] += time.perf_counter() - t; cnt["calls"] += 1
lib.call = call
for i in range(5):
    m.training_step(batch, i)
torch.cuda.synchronize()
acc.clear(); cnt.clear()
N = 10
t0 = time.perf_counter()
for i in range(N):
    m.training_step(batch, 5 + i)
tot = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"enqueue total {tot / N * 1e3:.2f} ms/step, {cnt['calls'] / N:.0f} C-ABI calls/step")
for k, v in acc.items():
    print(f"  {v / N * 1e3:7.2f} ms  {k}")
