"""Kernel profile driver for synthesise() (BASELINE config[4]: 64 sentences, eager decode): run under
rocprofv3 --kernel-trace --stats; `python tools/stats_per_step.py <csv> <reps>` prints ms per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision
from optispeech_amd.config import ModelConfig, make_optispeech
from optispeech_amd.values import InferenceInputs
precision.set_precision("bf16")
dev = "cuda"
torch.manual_seed(0)
m = make_optispeech(ModelConfig(), batch_size=32, pretraining_steps=0).to(dev).eval()
g = torch.Generator().manual_seed(7)
n = 64
x_len = torch.randint(64, 129, (n,), generator=g); x_len[0] = 128
x = torch.randint(1, 159, (n, 128), generator=g) * (torch.arange(128)[None] < x_len[:, None])
dur = torch.randint(4, 9, (n, 128), generator=g)
inp = InferenceInputs(clean_text="", x=x, x_lengths=x_len, d_factor=1.0, p_factor=1.0, e_factor=1.0)
REPS = int(os.environ.get("REPS", "10"))
for _ in range(3):
    o = m.synthesise(inp, durations_override=dur)
torch.cuda.synchronize()
for _ in range(REPS):
    o = m.synthesise(inp, durations_override=dur)
torch.cuda.synchronize()
print("rtf", o.rtf, "latency ms", o.latency, "am_rtf", o.am_rtf, "v_rtf", o.v_rtf)
