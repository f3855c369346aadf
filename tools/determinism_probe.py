"""Run-to-run reproducibility of the gradient arenas: the same step from the same state, several times in one process
(PREC=f32|bf16, REPS, B).  Run two copies at once to put the GPU under contention (the two-rank single-device test does)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optispeech_amd import precision, rng
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
from oracle import schema as S
precision.set_precision(os.environ.get("PREC", "bf16"))
c = S.SMALL
cfg = ModelConfig(dim=c.dim, enc_inter=c.enc_inter, dec_inter=c.dec_inter, dur=c.dur + (0.0,), pitch=c.pitch + (0.0,),
                  energy=c.energy + (0.0,), voc_dim=c.voc_dim, voc_inter=c.voc_inter, voc_layers=c.voc_layers).no_dropout()
B = int(os.environ.get("B", "2"))
batch = synthetic_batch(B, 24, 96, cfg, seed=50, ragged=True, device="cuda")
grads = []
names = None
for rep in range(int(os.environ.get("REPS", "6"))):
    torch.manual_seed(7); torch.cuda.manual_seed(7); rng.manual_seed(7, 0)
    m = make_optispeech(cfg, batch_size=B, pretraining_steps=0).to("cuda").train()
    m.generator.segment_rand01 = torch.tensor([0.25, 0.6], device="cuda")[:B]
    og, od = m.optimizers()
    m.training_step(batch, 0)
    torch.cuda.synchronize()
    grads.append((og.arena.grad.detach().clone(), od.arena.grad.detach().clone()))
    if names is None:
        names = []
        for o, pre in ((og, "G"), (od, "D")):
            by = {id(p): n for n, p in m.named_parameters()}
            names.append([(by[id(p)], off, p.numel()) for p, off in zip(o.arena.params, o.arena.offsets)])
tag = os.environ.get("TAG", "")
for i in range(1, len(grads)):
    for a, nm in ((0, "G"), (1, "D")):
        d = (grads[i][a] - grads[0][a])
        e = (d.norm() / grads[0][a].norm()).item()
        msg = f"{tag} run {i} vs 0: {nm} arena {e:.2e}"
        if e > 1e-4:
            worst = max(names[a], key=lambda t: d[t[1]:t[1] + t[2]].norm().item())
            sl = slice(worst[1], worst[1] + worst[2])
            msg += f"  worst {worst[0]} rel {(d[sl].norm() / grads[0][a][sl].norm().clamp_min(1e-20)).item():.2e}"
        print(msg)
