"""Serialised per-stage GPU+host wall time of the training step (device drained between the stages): which stage is slower when the
vocoder / acoustic model are taped?  Env: OSP_TAPE_AM / OSP_TAPE_VOC / OSP_TAPE_SEGMENTS."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import precision, rng, tape
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
torch.manual_seed(0); rng.manual_seed(0, 0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1, device="cuda")
m.optimizers()
for i in range(5):
    m.training_step(batch, i)
torch.cuda.synchronize()
acc = {}
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    acc[name + " (host)"] = acc.get(name + " (host)", 0.0) + th * 1e3
    return r
pb = m._process_batch
m._process_batch = lambda b: timed("  generator (_process_batch)", lambda: pb(b))
N = 8
for it in range(N):
    st = m._new_step_state(10 + it)
    rng.advance(); m._push_seed()
    timed("G forward (generator + disc forward_gen)", lambda: m._stage_g_forward(st, batch))
    timed("G backward", lambda: m._stage_g_backward(st))
    timed("D phase", lambda: m._stage_d(st, batch))
    timed("opt G", lambda: m._stage_opt_g(st))
    timed("opt D", lambda: m._stage_opt_d(st))
    rng.use_device_seed(None)
print(os.environ.get("TAG", ""), tape.stats())
for k, v in acc.items():
    print(f"{k:48s} {v / N:8.2f} ms")
if os.environ.get("PROFILE", "0") == "1":
    import cProfile, pstats
    torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    for it in range(6):
        st = m._new_step_state(50 + it)
        rng.advance(); m._push_seed()
        torch.cuda.synchronize()
        pr.enable()
        m._stage_g_forward(st, batch)
        pr.disable()
        torch.cuda.synchronize()
        m._stage_g_backward(st); m._stage_d(st, batch); m._stage_opt_g(st); m._stage_opt_d(st)
        rng.use_device_seed(None)
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
