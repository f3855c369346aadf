"""Steady-state training steps only (for rocprofv3 --kernel-trace --stats): STEPS steps after 3 warm-ups; divide the totals
by (STEPS + 3).  Env: STEPS (20), STREAMS (1), GRAPH (0), PRECISION (bf16)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision(os.environ.get("PRECISION", "bf16"))
torch.manual_seed(1234); rng.manual_seed(1234, 0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1234, device="cuda")
m.optimizers()
m.graph_steps = os.environ.get("GRAPH", "0") == "1"
n = int(os.environ.get("STEPS", "20"))
import time
m.pipeline_steps = os.environ.get("OSP_PIPELINE_STEPS", "0") == "1" or m.pipeline_steps
for i in range(3):
    m.training_step(batch, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3, 3 + n):
    m.training_step(batch, i)
torch.cuda.synchronize()
print("done", 3 + n, "steps;", round((time.perf_counter() - t0) / n * 1e3, 2), "ms per step over the last", n)
