#!/usr/bin/env python3
"""Per-phase wall/GPU timing of the training step (diagnostic; not the benchmark)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
from optispeech_amd import rng

from optispeech_amd import precision
precision.set_precision(os.environ.get("OSP_PRECISION", "f32"))
print("precision:", precision.get_precision())
dev = "cuda"
torch.manual_seed(0)
cfg = ModelConfig()
m = make_optispeech(cfg, pretraining_steps=0).to(dev).train()
batch = synthetic_batch(32, 128, 800, cfg, device=dev)
opt_g, opt_d = m.optimizers()
for i in range(3):
    m.training_step(batch, i)
torch.cuda.synchronize()

def timed(name, fn, acc):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r

acc = {}
N = 5
for it in range(N):
    rng.advance()
    for p in m.discriminator.parameters(): p.requires_grad_(False)
    go = timed("gen_forward(_process_batch)", lambda: m._process_batch(batch), acc)
    adv = timed("disc.forward_gen", lambda: m.discriminator.forward_gen(go["wav"], go["wav_hat"])[0], acc)
    loss = go["loss"] + adv
    opt_g.zero_grad()
    timed("G backward", lambda: loss.backward(), acc)
    for p in m.discriminator.parameters(): p.requires_grad_(True)
    timed("opt_g.step", lambda: opt_g.step(max_norm=10), acc)
    ld = timed("disc.forward_disc", lambda: m.discriminator.forward_disc(go["wav"], go["wav_hat"].detach())[0], acc)
    opt_d.zero_grad()
    timed("D backward", lambda: ld.backward(), acc)
    timed("opt_d.step", lambda: opt_d.step(max_norm=10), acc)
tot = 0
for k, v in acc.items():
    print(f"{k:32s} {v / N:8.2f} ms"); tot += v / N
print(f"{'sum':32s} {tot:8.2f} ms")

# finer: generator forward pieces under no_grad wall time incl. launch overhead
g = m.generator
with torch.no_grad():
    x = batch["x"]; xl = batch["x_lengths"]; ml = batch["mel_lengths"]
    acc2 = {}
    from optispeech_amd.model.generator import sequence_mask
    ipm = ~sequence_mask(xl, 128); tpm = ~sequence_mask(ml, 800)
    for it in range(N):
        h = timed("text_embedding", lambda: g.text_embedding(x)[0], acc2)
        h = timed("encoder", lambda: g.encoder(h, ipm), acc2)
        feats = batch["mel"].transpose(1, 2).contiguous()
        lp = timed("alignment_module", lambda: g.alignment_module(h, feats, xl, ml, ipm), acc2)
        from optispeech_amd.model.alignments import viterbi_decode, average_by_duration
        ds, path, bi = timed("viterbi(MAS)", lambda: viterbi_decode(lp, xl, ml), acc2)
        timed("duration_predictor", lambda: g.duration_predictor(h, ipm), acc2)
        pa, ea = timed("average_by_duration", lambda: average_by_duration(ds, batch["pitches"], batch["energies"], xl, ml), acc2)
        h2, _ = timed("pitch_predictor", lambda: g.pitch_predictor(h, ipm, pa), acc2)
        h3, _ = timed("energy_predictor", lambda: g.energy_predictor(h2, ipm, ea), acc2)
        y = timed("upsampler", lambda: g.feature_upsampler(h3, ds, xl, ml, 800), acc2)
        y = timed("decoder", lambda: g.decoder(y, tpm), acc2)
        seg = y[:, :64].contiguous()
        timed("vocoder(64 frames)", lambda: g.vocoder(seg), acc2)
        from optispeech_amd import kernels as K
        timed("forwardsum_ctc", lambda: K.forwardsum_ctc(lp, xl, ml), acc2)
    for k, v in acc2.items():
        print(f"  fwd {k:28s} {v / N:8.3f} ms")
