"""In-situ GPU timeline of the pipelined training step WITHOUT a profiler: HIP events recorded on the current stream at the
boundaries of the step's stages (no synchronisation inside the loop), read back after the run.  Shows, per step, when each stage starts
and ends on the device relative to the step's first event -- i.e. what overlaps what in the REAL schedule (under rocprofv3 the host is
slower and the schedule differs).  Env: the library's schedule knobs; NB (32)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision(os.environ.get("PRECISION", "bf16"))
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig()
NB = int(os.environ.get("NB", "32"))
m = make_optispeech(cfg, batch_size=NB, pretraining_steps=0).to("cuda").train()
m.pipeline_steps = True
batch = synthetic_batch(NB, 128, 800, cfg, seed=1, device="cuda")
m.optimizers()
marks = []          # (step, label, event)
cur = {"step": -1}


def mark(label):
    e = torch.cuda.Event(enable_timing=True)
    e.record()      # on the CURRENT stream
    marks.append((cur["step"], label, e))


def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        mark(label + " >")
        r = f(*a, **k)
        mark(label + " <")
        return r
    setattr(obj, name, g)


wrap(m, "_process_batch", "G fwd: acoustic model + vocoder")
wrap(m.discriminator, "forward_gen", "G fwd: discriminators on (real, generated) + losses")
wrap(m, "_stage_g_backward", "G backward")
wrap(m, "_stage_opt_g", "AdamW(G)")
wrap(m, "_stage_d", "D phase fwd + bwd [dstream]")
wrap(m, "_stage_opt_d", "AdamW(D) [dstream]")
for i in range(8):
    m.training_step(batch, i)
torch.cuda.synchronize()
marks.clear()
N = 12
t0 = time.perf_counter()
host = []
for i in range(N):
    cur["step"] = i
    th = time.perf_counter()
    m.training_step(batch, 8 + i)
    host.append((time.perf_counter() - th) * 1e3)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
print(f"{wall:.2f} ms per step (wall, {N} steps); host time inside training_step: " + " ".join(f"{h:.1f}" for h in host))
base = {}
for s, lab, e in marks:
    base.setdefault(s, e)
first = marks[0][2]
for s in range(4, N - 1):
    print(f"--- step {s}: events relative to the step's first event (ms); absolute start {first.elapsed_time(base[s]):.2f}")
    for s2, lab, e in marks:
        if s2 == s:
            print(f"    {base[s].elapsed_time(e):7.2f}  {lab}")
