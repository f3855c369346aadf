#!/usr/bin/env python3
"""Host (enqueue) cost of the training step on a tiny batch (the GPU is never the bound) + cProfile of 20 steps.
Env: BACKBONE=transformer for configs[3]; the schedule knobs of the library (OSP_TAPE_SEGMENTS, OSP_TAPES, ...), PROFILE=0 to skip cProfile."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import precision, rng, _lib
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig(backbone=os.environ["BACKBONE"]) if os.environ.get("BACKBONE") else ModelConfig()
m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
m.pipeline_steps = True
batch = synthetic_batch(2, 16, 72, cfg, seed=1, device="cuda")
m.optimizers()
for i in range(8):
    m.training_step(batch, i)
torch.cuda.synchronize()
import gc
for rep in range(3):
    t0 = time.perf_counter(); n = 30
    for i in range(n):
        m.training_step(batch, 10 + i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"host enqueue {th/n*1e3:.2f} ms/step (B=2: GPU work negligible)", flush=True)
lib = _lib.lib()
cnt = {"n": 0}
orig = lib.call
if os.environ.get("COUNT", "1") == "1":
    import collections
    names = collections.Counter()
    def call(name, *a):
        names[name] += 1
        return orig(name, *a)
    lib.call = call
    for i in range(5):
        m.training_step(batch, 200 + i)
    torch.cuda.synchronize()
    lib.call = orig
    print("direct C-ABI calls / step:", sum(names.values()) / 5)
    print("  ", ", ".join(f"{k[4:]} {v / 5:.0f}" for k, v in names.most_common(40)))
if os.environ.get("PROFILE", "1") == "1":
    import cProfile, pstats, io
    torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile(); pr.enable()
    N = 20
    for i in range(N):
        m.training_step(batch, 300 + i)
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s); st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(70)
    out = s.getvalue().replace(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/", "")
    print(f"(cProfile over {N} steps: divide by {N})")
    print(out)
