"""Find the first autograd Function of the step whose backward sees or produces a non-finite value (transformer, f32)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import precision, rng, tape, ops, kernels as K
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision(os.environ.get("PRECISION", "f32"))
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig(backbone=os.environ.get("BACKBONE", "transformer"))
m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(2, 24, 96, cfg, seed=5, device="cuda")
STEP = [0]
seen = set()

def fin(t):
    return (not isinstance(t, torch.Tensor)) or (not t.is_floating_point()) or bool(torch.isfinite(t).all())

def wrap(cls):
    ob = cls.backward
    def backward(ctx, *grads):
        torch.cuda.synchronize()
        bad_in = [i for i, g in enumerate(grads) if not fin(g)]
        bad_saved = [i for i, t in enumerate(getattr(ctx, "saved_tensors", ())) if not fin(t)]
        out = ob(ctx, *grads)
        torch.cuda.synchronize()
        outs = out if isinstance(out, tuple) else (out,)
        bad_out = [i for i, g in enumerate(outs) if not fin(g)]
        if (bad_in or bad_saved or bad_out) and (cls.__name__, STEP[0]) not in seen:
            seen.add((cls.__name__, STEP[0]))
            print(f"step {STEP[0]} {cls.__name__}.backward: non-finite grads in {bad_in} saved {bad_saved} out {bad_out}", flush=True)
        return out
    cls.backward = staticmethod(backward)

for name in dir(ops):
    c = getattr(ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)
for i in range(4):
    STEP[0] = i
    m.training_step(batch, i)
    torch.cuda.synchronize()
    bad = [k for k, p in m.generator.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("step", i, "non-finite parameter gradients:", bad, flush=True)
