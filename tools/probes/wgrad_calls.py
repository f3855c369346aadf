"""Every weight-gradient entry-point call of one B = 32 training step: shape, operand types, the symbol the dispatcher chose, and its
time with the device drained around it (tapes, side streams and sub-discriminator streams off)."""
import os, sys, ctypes, collections
os.environ.update(OSP_TAPES="0", OSP_WGRAD_STREAM="0", OSP_DISC_STREAMS="0", OSP_VOC_STREAM="0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import _lib, precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision("bf16")
torch.manual_seed(1234); rng.manual_seed(1234, 0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
batch = synthetic_batch(32, 128, 800, cfg, seed=1234, device="cuda")
m.optimizers()
for i in range(3):
    m.training_step(batch, i)
torch.cuda.synchronize()
lib = _lib.lib(); orig = lib.call
note = lib.cdll.osp_kernel_note_host
note.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
buf, fl = ctypes.create_string_buffer(128), ctypes.c_double(0.0)
agg = collections.OrderedDict()
def call(name, *a):
    if "wgrad" not in name:
        return orig(name, *a)
    torch.cuda.synchronize()
    note(buf, 128, ctypes.byref(fl))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *a); e1.record(); torch.cuda.synchronize()
    note(buf, 128, ctypes.byref(fl))
    ints = tuple(x for x in a if isinstance(x, int))[:14]
    k = (name, ints, buf.value.decode())
    v = agg.setdefault(k, [0, 0.0]); v[0] += 1; v[1] += e0.elapsed_time(e1)
lib.call = call
N = 2
for i in range(N):
    m.training_step(batch, 3 + i)
torch.cuda.synchronize()
tot = 0.0
for (name, ints, sym), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms / N
    print(f"{ms / N * 1e3:8.1f} us/step  x{n / N:4.1f}  avg {ms / n * 1e3:7.1f} us  {sym:36s} {name[4:]} {ints}")
print(f"total {tot:.2f} ms/step")
