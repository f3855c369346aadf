"""Where does a hipGraph replay of the training step spend its time?  host launch cost vs device time, relaunch blocking."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
from optispeech_amd.graphs import StepGraphs
from optispeech_amd import _lib

precision.set_precision("bf16")
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device="cuda")
m.optimizers()
m.graph_force_segments = os.environ.get("SEG", "0") == "1"
# count launches of one eager step
lib = _lib.lib(); orig = lib.call; n = [0]
def call(name, *a):
    n[0] += 1; orig(name, *a)
lib.call = call
for i in range(3):
    m.training_step(batch, i)
n[0] = 0
m.training_step(batch, 3)
torch.cuda.synchronize()
print("C-ABI calls per eager step:", n[0])
lib.call = orig
t0 = time.perf_counter()
for i in range(10):
    m.training_step(batch, i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"eager serial: enqueue {(t1-t0)/10*1e3:.2f} ms/step, drained {(t2-t0)/10*1e3:.2f} ms/step")
sg = StepGraphs(m, batch, warmup=2)
print("graphs:", len(sg.graphs))
torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sg.replay(batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"idle-start replay: host {(t1-t0)*1e3:.2f} ms, to completion {(t2-t0)*1e3:.2f} ms")
torch.cuda.synchronize()
t0 = time.perf_counter(); hs = []
for i in range(20):
    a = time.perf_counter(); sg.replay(batch); hs.append((time.perf_counter() - a) * 1e3)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"20 back-to-back replays: host {(t1-t0)/20*1e3:.2f} ms/step, drained {(t2-t0)/20*1e3:.2f} ms/step; per-replay host ms: {[round(h,1) for h in hs]}")
# raw graph launch only (no batch copy / scalars)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    for g in sg.graphs:
        g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"raw g.replay(): host {(t1-t0)/10*1e3:.2f} ms/step, drained {(t2-t0)/10*1e3:.2f}")
# device time of one replay by events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); sg.graphs[0].replay(); e1.record(); torch.cuda.synchronize()
print("event time of one replay of graph 0:", e0.elapsed_time(e1), "ms")
try:
    sg.graphs[0].enable_debug_mode()
except Exception as e:
    print("debug mode n/a", e)
