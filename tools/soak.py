"""Soak of the default training configuration (bf16, multi-stream lanes, pipelined steps, call tapes): STEPS consecutive GAN steps on
the fixed synthetic batch; ms / step per block of 250 and the logged losses at the block ends (finite, moving)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optispeech_amd import precision, rng, tape
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision(os.environ.get("PRECISION", "bf16"))          # PRECISION=mixed: the parity mode
torch.manual_seed(1234); rng.manual_seed(1234, 0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1234, device="cuda")
m.optimizers()
m.pipeline_steps = True
n, blk = int(os.environ.get("STEPS", "1500")), 250
for i in range(5):
    m.training_step(batch, i)
torch.cuda.synchronize()
for b0 in range(5, 5 + n, blk):
    t0 = time.perf_counter()
    for i in range(b0, b0 + blk):
        m.training_step(batch, i)
    m.join(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / blk * 1e3
    logs = m.fetch_logs()
    keys = ("total_loss/train_am_loss", "total_loss/train_gen_adv_loss", "total_loss/discriminator")
    vals = {k.split("/")[-1]: round(float(logs[k]), 3) for k in keys if k in logs}
    assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
    print(f"steps {b0 - 5:5d}-{b0 - 5 + blk - 1:5d}: {dt:6.2f} ms/step  {vals}", flush=True)
print("tapes:", tape.stats())
