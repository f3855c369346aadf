#!/usr/bin/env python3
"""synthesise() (configs[4]: 64 sentences) eager / captured decode / captured encode + decode, same box: latency and RTF (median of REPS)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import precision
from optispeech_amd.config import ModelConfig, make_optispeech
from optispeech_amd.values import InferenceInputs
precision.set_precision("bf16")
torch.manual_seed(0)
m = make_optispeech(ModelConfig(), batch_size=32, pretraining_steps=0).to("cuda").eval()
g = torch.Generator().manual_seed(7)
n = 64
x_len = torch.randint(64, 129, (n,), generator=g); x_len[0] = 128
x = torch.randint(1, 159, (n, 128), generator=g) * (torch.arange(128)[None] < x_len[:, None])
dur = torch.randint(4, 9, (n, 128), generator=g)
inp = InferenceInputs(clean_text="", x=x, x_lengths=x_len, d_factor=1.0, p_factor=1.0, e_factor=1.0)
REPS = int(os.environ.get("REPS", "15"))
ref = None
for name, gd, ge in (("eager", False, False), ("captured decode", True, False), ("captured encode + decode", True, True)):
    m.generator.graph_decode, m.generator.graph_encode = gd, ge
    for _ in range(3):
        o = m.synthesise(inp, durations_override=dur)
    outs = [m.synthesise(inp, durations_override=dur) for _ in range(REPS)]
    lat = statistics.median(o.latency for o in outs); rtf = statistics.median(o.rtf for o in outs); am = statistics.median(o.am_rtf for o in outs)
    same = True if ref is None else bool(torch.equal(ref.wav, outs[-1].wav) and torch.equal(ref.durations, outs[-1].durations))
    ref = ref or outs[-1]
    print(f"{name:26s} latency {lat:6.3f} ms  rtf {rtf:.3e}  am_rtf {am:.3e}  v_rtf {rtf - am:.3e}  output == eager: {same}")
