"""Does a taped generator segment replayed on a DIFFERENT ragged batch (same padded shapes) equal the eager step on that batch?
Two steps without an optimizer update on batch 0 then batch 1: gradients and logged losses of the second step, tapes on vs off."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import schema as S
from optispeech_amd import precision, rng, tape
from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
precision.set_precision(os.environ.get("PREC", "f32"))


def run(tapes):
    tape.ENABLED = tapes
    c = S.SMALL
    cfg = ModelConfig(dim=c.dim, enc_inter=c.enc_inter, dec_inter=c.dec_inter, dur=c.dur + (0.0,), pitch=c.pitch + (0.0,),
                      energy=c.energy + (0.0,), voc_dim=c.voc_dim, voc_inter=c.voc_inter, voc_layers=c.voc_layers).no_dropout()
    torch.manual_seed(7); rng.manual_seed(7, 0)
    m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
    og, od = m.optimizers()
    got = {}
    for name, o in (("g", og), ("d", od)):
        o.step = (lambda n, oo: (lambda *a, **k: got.__setitem__(n, oo.arena.grad.detach().clone())))(name, o)
    out = []
    for r in (0, 1, 1, 0):
        b = synthetic_batch(2, 24, 96, cfg, seed=50 + r, ragged=True, device="cuda")
        m.generator.segment_rand01 = torch.tensor([0.25 + 0.5 * r, 0.6 - 0.3 * r], device="cuda")
        m.training_step(b, 0)
        logs = m.fetch_logs()
        torch.cuda.synchronize()
        out.append((got["g"].clone(), got["d"].clone(), dict(logs), m))
    return out


a = run(False)
b = run(True)
print(tape.stats())
for i, ((ga, da, la, m), (gb, db, lb, _)) in enumerate(zip(a, b)):
    print(f"step {i}: G grads rel diff {((ga - gb).norm() / ga.norm()).item():.2e}, D grads {((da - db).norm() / da.norm()).item():.2e}")
    for k in la:
        if abs(la[k] - lb[k]) > 1e-5 * abs(la[k]) + 1e-7:
            print(f"     log {k}: eager {la[k]:.6f} taped {lb[k]:.6f}")
    if ((ga - gb).norm() / ga.norm()).item() > 1e-4:
        o = m.optimizers()[0]
        by = {id(p): n for n, p in m.named_parameters()}
        worst = sorted(((((ga - gb)[off:off + p.numel()]).norm().item() / (ga[off:off + p.numel()].norm().item() + 1e-12), by[id(p)]) for p, off in zip(o.arena.params, o.arena.offsets)), reverse=True)[:12]
        print("     worst params:", ", ".join(f"{n} {v:.1e}" for v, n in worst))
