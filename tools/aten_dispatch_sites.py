#!/usr/bin/env python3
"""Which ATen ops does one training step still dispatch, and from where?  A TorchDispatchMode counts every op (view / metadata ops
excluded) of step 5 by (op, first frame inside optispeech_amd/ -- 'autograd engine' when there is none: gradient accumulation)."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch

NO_KERNEL = ("view", "reshape", "_unsafe_view", "expand", "permute", "transpose", "t.default", "squeeze", "unsqueeze", "slice", "select",
             "detach", "alias", "as_strided", "empty", "_local_scalar", "is_", "size", "stride", "numel", "narrow", "split", "unbind",
             "record_stream", "_to_copy.default_meta", "lift_fresh", "chunk", "resize_", "set_")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in NO_KERNEL):
            site = "autograd engine"
            for f in reversed(traceback.extract_stack()[:-1]):
                if "optispeech_amd/" in f.filename:
                    site = f"{f.filename.split('optispeech_amd/')[-1]}:{f.lineno}"
                    break
            self.n[(name.replace("aten.", ""), site)] += 1
        return func(*args, **(kwargs or {}))


precision.set_precision("bf16")
torch.manual_seed(0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
m.pipeline_steps = True
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device="cuda")
m.optimizers()
torch.autograd.set_multithreading_enabled(False)
for i in range(5):
    m.training_step(batch, i)
torch.cuda.synchronize()
log = Log()
with log:
    m.training_step(batch, 5)
torch.cuda.synchronize()
tot = sum(log.n.values())
print("ATen ops dispatched in one step (views excluded):", tot)
for (op, site), c in log.n.most_common(60):
    print(f"{c:5d}  {op:34s} {site}")
