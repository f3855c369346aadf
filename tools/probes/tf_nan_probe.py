"""Which of the round-4 element-wise entry points makes a parameter gradient non-finite?  BISECT=<name> swaps one back to torch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import precision, rng, tape, kernels as K
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
from optispeech_amd.model import generator as G
b = os.environ.get("BISECT", "")
if b == "scale_dev": K.ew_scale_dev = lambda x, s, c=1.0, out=None: x * s * c
if b == "mul_rows": K.ew_mul_rows = lambda x, r, out=None: x * r.reshape(-1)[:, None]
if b == "relu_mask": K.relu_mask = lambda g, y: g * (y > 0)
if b == "axpby": K.ew_axpby = lambda x, y, a=1.0, b=1.0, out=None: (a * x + (b * y if y is not None else b))
if b == "sum_scaled": K.sum_scaled = lambda x, scale: x.sum() * scale
if b == "dot":
    K.dot_multi = lambda ts, cs: torch.dot(torch.stack(ts), torch.tensor(cs, device=ts[0].device))
    K.scale_vec = lambda g, cs: g * torch.tensor(cs, device=g.device)
if b == "masks": G.padding_mask = lambda l, T: ~G.sequence_mask(l, T)
if b == "transpose": K.transpose_last2 = lambda x: x.transpose(1, 2).contiguous()
precision.set_precision(os.environ.get("PRECISION", "f32"))
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig(backbone=os.environ.get("BACKBONE", "transformer"))
m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(2, 24, 96, cfg, seed=5, device="cuda")
first = None
for i in range(4):
    m.training_step(batch, i)
    torch.cuda.synchronize()
    bad = [k for k, p in m.generator.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    nanw = [k for k, v in m.state_dict().items() if v.is_floating_point() and not torch.isfinite(v).all()]
    if (bad or nanw) and first is None:
        first = (i, bad[:6], nanw[:3])
print("BISECT", b or "-", "first non-finite:", first)
