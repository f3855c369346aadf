#!/usr/bin/env python3
"""Host-side profile of the training step (the step is CPU-enqueue-bound): cProfile by own time + per-entry-point call counts."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision, _lib
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
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
import cProfile, pstats
torch.autograd.set_multithreading_enabled(False)      # backward on this thread, so that the profiler sees it
pr = cProfile.Profile(); pr.enable()
N = 10
for i in range(N):
    m.training_step(batch, 30 + i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(60)
