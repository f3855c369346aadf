#!/usr/bin/env python3
"""Is the training step CPU-launch-bound?  CPU time to enqueue N steps vs wall time until the GPU drained them (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
dev = "cuda"
torch.manual_seed(0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to(dev).train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device=dev)
m.optimizers()
for i in range(5):
    m.training_step(batch, i)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N):
    m.training_step(batch, 5 + i)
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {t_cpu / N * 1e3:.2f} ms/step   drained {t_all / N * 1e3:.2f} ms/step   (CPU-bound if the two are equal)")
ts = []
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.training_step(batch, 100 + i)
    ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
print("single-step enqueue on an empty queue (ms):", [round(t, 1) for t in ts])
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5):
    m.training_step(batch, 30 + i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(25)
