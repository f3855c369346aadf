#!/usr/bin/env python3
"""Which torch (aten) ops does a training step still issue, how often, and what do they cost on the host (diagnostic)."""
import os, sys
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
m.pipeline_steps = True
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device=dev)
m.optimizers()
for i in range(5):
    m.training_step(batch, i)
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(N):
        m.training_step(batch, 5 + i)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in rows)
print(f"self CPU total {tot / N / 1e3:.2f} ms/step (profiler overhead included)")
for e in rows[:45]:
    print(f"{e.self_cpu_time_total / N / 1e3:7.3f} ms/step  x{e.count / N:7.1f}  {e.key[:90]}")
