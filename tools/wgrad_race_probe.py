"""Repeat the training step of ONE model on two alternating micro-batches without updating the weights and compare every step's
gradient arenas with the first step on the same batch: run-to-run differences above the f32-atomic noise (1e-6) mean a race.
usage: python tools/wgrad_race_probe.py [f32|bf16] [steps]"""
import sys
import torch

sys.path.insert(0, ".")
from tests.test_gpu_dp import _build, _batches, _grads_of_one_step      # noqa: E402
from optispeech_amd import precision                                     # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
precision.set_precision(mode)
cfg, m = _build(7)
batches = _batches(cfg)
m.optimizers()
names = {}
for o, tag in zip(m.optimizers(), "gd"):
    by = {id(p): n for n, p in m.named_parameters()}
    names[tag] = [(by[id(p)], off, p.numel()) for p, off in zip(o.arena.params, o.arena.offsets)]
first = {}
bad = 0
for s in range(steps):
    k = s % 2
    g, d = _grads_of_one_step(m, *batches[k])
    if k not in first:
        first[k] = (g, d)
        continue
    for tag, got, ref in (("g", g, first[k][0]), ("d", d, first[k][1])):
        e = ((got - ref).norm() / ref.norm()).item()
        if e > 2e-5:
            bad += 1
            worst = sorted((((got - ref)[off:off + n].norm().item(), nm) for nm, off, n in names[tag]), reverse=True)[:3]
            print(f"step {s} batch {k} {tag}: rel {e:.2e}  " + ", ".join(f"{nm} {v:.1e}" for v, nm in worst), flush=True)
print(f"{mode}: {bad} deviating arena(s) in {steps - 2} compared steps")
