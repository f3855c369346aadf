import os, sys, torch
sys.path.insert(0, "/root/repo")
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig(backbone="transformer")
m = make_optispeech(cfg, batch_size=4, pretraining_steps=0).to("cuda").train()
m.pipeline_steps = True
b = synthetic_batch(4, 32, 160, cfg, seed=1, device="cuda")
m.optimizers()
for i in range(3):
    m.training_step(b, i)
torch.cuda.synchronize()
