#!/usr/bin/env python3
"""Shapes of the aten::copy_ / fill_ / zero_ / add calls of a training step (which tensors does the torch glue still touch)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
from torch.profiler import profile, ProfilerActivity
precision.set_precision("bf16")
dev = "cuda"
torch.manual_seed(0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to(dev).train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device=dev)
m.optimizers()
for i in range(4):
    m.training_step(batch, i)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    for i in range(N):
        m.training_step(batch, 5 + i)
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::_to_copy", "aten::clone", "aten::zeros"):
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        gp = e.cpu_parent.cpu_parent.name if (e.cpu_parent is not None and e.cpu_parent.cpu_parent is not None) else "-"
        agg[(e.name, str(e.input_shapes)[:70], par[:40], gp[:40])] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c / N:6.1f}  {k[0]:14s} {k[1]:72s} <- {k[2]} <- {k[3]}")
