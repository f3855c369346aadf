"""Pure host (enqueue) cost of the training step: the same launch sequence on a tiny batch, where the GPU is never the bound."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
m.pipeline_steps = True
batch = synthetic_batch(2, 16, 72, cfg, seed=1, device="cuda")
m.optimizers()
for i in range(6):
    m.training_step(batch, i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); n = 20
    for i in range(n):
        m.training_step(batch, 10 + i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print(f"host enqueue {th/n*1e3:.2f} ms/step, wall {tt/n*1e3:.2f} ms/step (B=2: GPU work negligible)")
import gc
gc.collect(); gc.freeze(); gc.disable()
for rep in range(3):
    t0 = time.perf_counter(); n = 20
    for i in range(n):
        m.training_step(batch, 100 + i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"gc off: host enqueue {th/n*1e3:.2f} ms/step")
