#!/usr/bin/env python3
"""Aggregate GPU time of every C-ABI call by (entry point, shape signature) over a few training steps (diagnostic)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import _lib, precision
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision(os.environ.get("OSP_PRECISION", "bf16"))
dev = "cuda"
torch.manual_seed(0)
cfg = ModelConfig()
m = make_optispeech(cfg, pretraining_steps=0).to(dev).train()
batch = synthetic_batch(32, 128, 800, cfg, device=dev)
for i in range(3):
    m.training_step(batch, i)
torch.cuda.synchronize()
lib = _lib.lib(); orig = lib.call
events = []
def sig(name, args):
    ints = [a for a in args if isinstance(a, int) and not isinstance(a, bool)]
    if name == "osp_conv_gemm_bf16": return (name, "M", args[3], "Tr", args[4], "Cin", args[6], "taps", args[7], "N", args[17], "astep", args[8])
    if name == "osp_conv2d_gemm_bf16": return (name, "M", args[3], "Cin", args[8], "taps", args[9], "N", args[23])
    if name == "osp_conv2d_dgrad_bf16":
        return (name, "U", args[6], "H", args[7], "W", args[8], "Cin", args[11], "Cout", args[12], "K", (args[13], args[14]), "s", (args[15], args[16]))
    if name == "osp_conv_gemm_f32": return (name, "M", args[2], "Cin", args[4], "taps", args[5], "N", args[12], "epi", args[15])
    if name == "osp_conv_wgrad_bf16": return (name, "M", args[6], "N", args[9], "Cin", args[10], "taps", args[11])
    if name == "osp_conv2d_wgrad_bf16": return (name, "M", args[6], "N", args[11], "Cin", args[12], "taps", args[13])
    if name == "osp_conv_wgrad_f32": return (name, "M", args[4], "N", args[6], "Cin", args[7], "taps", args[8])
    return (name,) + tuple(ints[:4])
SYNC = os.environ.get("SYNC", "0") == "1"      # drain the stream before every call: kernel time without launch gaps
def call(name, *args):
    if SYNC:
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *args); e1.record()
    events.append((sig(name, args), e0, e1))
lib.call = call
N = 3
for i in range(N):
    m.training_step(batch, 3 + i)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for k, a, b in events:
    agg[k][0] += a.elapsed_time(b); agg[k][1] += 1
tot = sum(v[0] for v in agg.values()) / N
print(f"osp kernels: {tot:.2f} ms/step in {len(events)/N:.0f} calls/step")
byname = collections.defaultdict(lambda: [0.0, 0])
for k, v in agg.items():
    byname[k[0]][0] += v[0]; byname[k[0]][1] += v[1]
for k, v in sorted(byname.items(), key=lambda kv: -kv[1][0]):
    print(f"  {v[0]/N:8.3f} ms/step  x{v[1]/N:6.1f}  {k}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", "30"))]:
    tf = ""
    if k[0] == "osp_conv2d_gemm_bf16":
        tf = f"{2.0*k[2]*k[4]*k[6]*k[8]/(v[0]/v[1]*1e-3)/1e12:6.0f} TF"
    if k[0] == "osp_conv2d_dgrad_bf16":
        U, H, W, Cin, Cout, (KH, KW), (sh, sw) = k[2], k[4], k[6], k[8], k[10], k[12], k[14]
        tf = f"{2.0*U*H*W*Cin*Cout*KH*KW/(sh*sw)/(v[0]/v[1]*1e-3)/1e12:6.0f} TF"
    if k[0] == "osp_conv2d_wgrad_bf16":
        tf = f"{2.0*k[2]*k[4]*k[6]*k[8]/(v[0]/v[1]*1e-3)/1e12:6.0f} TF"
    print(f"{v[0]/N:8.3f} ms/step  x{v[1]/N:5.1f}  avg {v[0]/v[1]*1e3:8.1f} us {tf}  {k}")
