"""Run-to-run reproducibility of the gradient arenas: the same step from the same state, several times in one process."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
cfg = ModelConfig()
B = int(os.environ.get("B", "4"))
batch = synthetic_batch(B, 32, 160, cfg, seed=1, device="cuda")
grads = []
for rep in range(4):
    torch.manual_seed(0); torch.cuda.manual_seed(0); rng.manual_seed(0, 0)
    m = make_optispeech(cfg, batch_size=B, pretraining_steps=0).to("cuda").train()
    m.generator.segment_rand01 = torch.rand(B, generator=torch.Generator().manual_seed(1)).cuda()
    og, od = m.optimizers()
    m.training_step(batch, 0)
    torch.cuda.synchronize()
    grads.append((og.arena.grad.detach().clone(), od.arena.grad.detach().clone()))
for i in range(1, 4):
    eg = ((grads[i][0] - grads[0][0]).norm() / grads[0][0].norm()).item()
    ed = ((grads[i][1] - grads[0][1]).norm() / grads[0][1].norm()).item()
    print(f"run {i} vs run 0: generator arena {eg:.2e}, discriminator arena {ed:.2e}")
