"""GPU-only time of the eager multi-stream step: block the GPU with a spin kernel, enqueue one full step behind it, measure
from the spin's end to the step's end with events (host enqueue time is then hidden)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device="cuda")
m.optimizers()
m.pipeline_steps = os.environ.get("PIPE", "0") == "1"
for i in range(5):
    m.training_step(batch, i)
torch.cuda.synchronize()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2.0e8))            # ~80-100 ms of spinning on the calling stream
    e0.record()
    t0 = time.perf_counter()
    n = 2
    for i in range(n):
        m.training_step(batch, 10 + i)
    m.join()
    e1.record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"host enqueue {th/n*1e3:.1f} ms/step; GPU time behind the spin {e0.elapsed_time(e1)/n:.2f} ms/step (pipeline={m.pipeline_steps})")
