#!/usr/bin/env python3
"""Per-shape timing of the degenerate-shape kernels (conv_rowdot_bf16_kernel: N = 1, conv_outer_bf16_kernel: Cin = 1) at the shapes a
B = 32 step really launches them on: every such call of one eager step is re-issued REP times back to back between two events
(same arguments, buffers still live), so the figure is kernel time without launch gaps.  SYMS=a,b selects other symbols -- but only
calls issued on torch's CURRENT stream are timed correctly: the events are recorded there, and the weight-gradient entry points
launch on their own side streams (their figures come out as launch overhead, ~4 us).
    OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 PYTHONPATH=. python tools/probes/degen_probe.py"""
import collections, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import _lib, precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch

REP = int(os.environ.get("REP", "30"))
WANT = tuple(os.environ.get("SYMS", "conv_rowdot_bf16_kernel,conv_outer_bf16_kernel").split(","))
precision.set_precision("bf16")
torch.manual_seed(1234); rng.manual_seed(1234, 0)
cfg = ModelConfig()
m = make_optispeech(cfg, batch_size=32, pretraining_steps=0).to("cuda").train()
m.tape_segments = False
batch = synthetic_batch(32, 128, 800, cfg, seed=1234, device="cuda")
m.optimizers()
for i in range(3):
    m.training_step(batch, i)
torch.cuda.synchronize()
lib = _lib.lib(); orig = lib.call
note = lib.cdll.osp_kernel_note_host
note.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
note_bytes = lib.cdll.osp_kernel_note_bytes_host
note_bytes.argtypes = [ctypes.POINTER(ctypes.c_double)]
buf, fl, by = ctypes.create_string_buffer(128), ctypes.c_double(0.0), ctypes.c_double(0.0)
rows = []


def ints(args):
    return tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 24))


def call(name, *args):
    if _lib._RECORD[0] is not None or not name.startswith("osp_conv"):
        return orig(name, *args)
    note(buf, 128, ctypes.byref(fl)); note_bytes(ctypes.byref(by))
    r = orig(name, *args)
    note(buf, 128, ctypes.byref(fl)); note_bytes(ctypes.byref(by))
    sym = buf.value.decode()
    if sym in WANT:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            orig(name, *args)
        e1.record(); torch.cuda.synchronize()
        note(buf, 128, ctypes.byref(fl)); note_bytes(ctypes.byref(by))
        rows.append((sym, name, ints(args), e0.elapsed_time(e1) * 1e3 / REP, by.value / REP))
    return r


lib.call = call
m.training_step(batch, 3)
torch.cuda.synchronize()
lib.call = orig
agg = collections.OrderedDict()
for sym, name, shape, us, b in rows:
    v = agg.setdefault((sym, name, shape), [0, 0.0, b])
    v[0] += 1; v[1] += us
tot = collections.Counter()
for (sym, name, shape), (n, us, b) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot[sym] += us
    print(f"{sym:28s} {name:24s} x{n}  {us / n:7.1f} us  {b / (us / n) / 1e3:7.0f} GB/s  {b / 1e6:7.2f} MB  {shape}")
for s, us in tot.items():
    print(f"total {s}: {us / 1e3:.3f} ms / step")
