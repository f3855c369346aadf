#!/usr/bin/env python3
"""Python call sites of the torch glue a training step still issues (zeros / cat / stack / .to / .contiguous / .clone ...): the
functions are wrapped with a counter keyed by the first frame inside optispeech_amd/."""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
dev = "cuda"
torch.manual_seed(0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to(dev).train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device=dev)
m.optimizers()
torch.autograd.set_multithreading_enabled(False)
for i in range(4):
    m.training_step(batch, i)
torch.cuda.synchronize()
agg = collections.Counter()
ON = [False]
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "optispeech_amd" in f.filename:
            return f"{f.filename.split('optispeech_amd/')[-1]}:{f.lineno} {f.line[:70] if f.line else ''}"
    return "?"
def wrap(obj, name, label, cond=None):
    fn = getattr(obj, name)
    def g(*a, **k):
        r = fn(*a, **k)
        if ON[0] and (cond is None or cond(a, k, r)):
            agg[(label, site())] += 1
        return r
    setattr(obj, name, g)
for nm in ("zeros", "zeros_like", "cat", "stack", "ones", "full", "arange", "tensor", "empty_like", "where", "sqrt", "clip", "clamp"):
    wrap(torch, nm, "torch." + nm)
T = torch.Tensor
wrap(T, "to", "Tensor.to", lambda a, k, r: r is not a[0])
wrap(T, "contiguous", "Tensor.contiguous", lambda a, k, r: r is not a[0] and r.data_ptr() != a[0].data_ptr())
wrap(T, "clone", "Tensor.clone")
wrap(T, "float", "Tensor.float", lambda a, k, r: r is not a[0])
wrap(T, "zero_", "Tensor.zero_")
wrap(T, "fill_", "Tensor.fill_")
wrap(T, "copy_", "Tensor.copy_")
wrap(T, "__add__", "Tensor.+")
wrap(T, "__mul__", "Tensor.*")
wrap(T, "__truediv__", "Tensor./")
wrap(T, "__sub__", "Tensor.-")
wrap(T, "sum", "Tensor.sum")
wrap(T, "mean", "Tensor.mean")
wrap(T, "reshape", "Tensor.reshape(copy)", lambda a, k, r: r.data_ptr() != a[0].data_ptr())
ON[0] = True
N = 2
for i in range(N):
    m.training_step(batch, 5 + i)
torch.cuda.synchronize()
ON[0] = False
tot = sum(agg.values()) / N
print(f"{tot:.0f} wrapped torch calls / step")
for (name, s), c in sorted(agg.items(), key=lambda kv: -kv[1])[:80]:
    print(f"{c / N:6.1f}  {name:22s} {s}")
