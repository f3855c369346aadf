#!/usr/bin/env python3
"""Per-shape inventory of the MFMA entry points over steady-state training steps: launches/step, average time with the
device drained before every call (kernel time without overlap), TFLOP/s.  Run with OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
P = {  # entry -> (names of the int arguments that make the shape, flops(args dict))
    "osp_conv_gemm_bf16": ("A a_bf16 lda M Trows Tin Cin taps a_step a_tapstep a_off a_rowscale B b_bf16 sBn sBtap sBk N C c_bf16 ldc Tc c_step c_off epi "
                           "bias gamma res res_bf16 ldr rowmask rowscale aux_out aux_in aux_bf16 ld_aux slope batch").split(),
    "osp_conv2d_gemm_bf16": "A a_bf16 lda M Trows Wrows Hin Win Cin taps KW a_step_h a_tapstep_h a_off_h a_step a_tapstep a_off B b_bf16 sBn sBtap_h sBtap sBk N".split(),
    "osp_conv2d_dgrad_bf16": "dy dy_bf16 wt w_bf16 dx dx_bf16 U H W Ho Wo Cin Cout KH KW sh sw".split(),
    "osp_conv_wgrad_bf16": "dY y_bf16 ldy X x_bf16 ldx M Trows Tin N Cin taps pad x_step arow oscale dW ldw db batch".split(),
    "osp_conv2d_wgrad_bf16": "dY y_bf16 ldy X x_bf16 ldx M Trows Wrows Hin Win N Cin taps".split(),
}
def key(name, args):
    d = dict(zip(P[name], args))
    if name == "osp_conv_gemm_bf16":
        b = max(1, d["batch"]); return (name, d["M"], d["N"], d["Cin"], d["taps"], b, d["a_bf16"], d["c_bf16"], d["epi"]), 2.0 * d["M"] * d["N"] * d["Cin"] * d["taps"] * b
    if name == "osp_conv2d_gemm_bf16":
        return (name, d["M"], d["N"], d["Cin"], d["taps"]), 2.0 * d["M"] * d["N"] * d["Cin"] * d["taps"]
    if name == "osp_conv2d_dgrad_bf16":
        return (name, d["U"] * d["H"] * d["W"], d["Cin"], d["Cout"], d["KH"] * d["KW"], d["sh"] * d["sw"]), \
            2.0 * d["U"] * d["H"] * d["W"] * d["Cin"] * d["Cout"] * d["KH"] * d["KW"] / (d["sh"] * d["sw"])
    if name == "osp_conv_wgrad_bf16":
        b = max(1, d["batch"]); return (name, d["M"], d["N"], d["Cin"], d["taps"], b), 2.0 * d["M"] * d["N"] * d["Cin"] * d["taps"] * b
    return (name, d["M"], d["N"], d["Cin"], d["taps"]), 2.0 * d["M"] * d["N"] * d["Cin"] * d["taps"]
events = []
where = {}
def call(name, *args):
    if name not in P:
        return orig(name, *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *args); e1.record()
    k, fl = key(name, args)
    if k not in where:
        import traceback
        fr = [f for f in traceback.extract_stack()[:-1] if "tools/" not in f.filename and "_lib" not in f.filename]
        where[k] = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-4:][::-1])
    events.append((k, fl, e0, e1))
lib.call = call
N = 3
for i in range(N):
    m.training_step(batch, 3 + i)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for k, fl, a, b in events:
    v = agg[k]; v[0] += a.elapsed_time(b); v[1] += 1; v[2] += fl
tot = sum(v[0] for v in agg.values()) / N
totfl = sum(v[2] for v in agg.values()) / N
print(f"MFMA entries: {tot:.2f} ms/step, {len(events)/N:.0f} calls/step, {totfl/1e12:.2f} TFLOP/step -> {totfl/tot/1e9:.0f} TF average")
acc = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", "60"))]:
    acc += v[0] / N
    print(f"{v[0]/N:7.3f} ms x{v[1]/N:5.1f} avg {v[0]/v[1]*1e3:7.1f} us {v[2]/v[0]/1e9:5.0f} TF  cum {acc:6.2f}  {k}  [{where[k]}]")
