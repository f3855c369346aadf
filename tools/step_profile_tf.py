"""tools/step_profile.py for the Transformer backbone variant (BASELINE configs[3])."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
torch.manual_seed(1234); rng.manual_seed(1234, 0)
cfg = ModelConfig(backbone="transformer")
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1234, device="cuda")
m.optimizers()
m.pipeline_steps = os.environ.get("PIPELINE", "1") == "1"
n = int(os.environ.get("STEPS", "10"))
import time
for i in range(3):
    m.training_step(batch, i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n):
    m.training_step(batch, 3 + i)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / n * 1e3)
