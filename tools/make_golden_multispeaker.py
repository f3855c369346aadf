#!/usr/bin/env python3
"""Golden fixture for the multi-speaker / multi-language path (sids / lids -> sid_embed / lid_embed added to the encoder
output, generator/__init__.py:62-65,112-117,235-246) by RUNNING THE REFERENCE generator here.

    python tools/make_golden_multispeaker.py       # writes tests/golden/gen_small_multispk.npz

Same recipe as tools/make_golden.py (stubs, MAS shim, injected segment starts; weights = oracle.schema.make_weights, not
stored) with num_speakers = 3, num_languages = 2: forward + backward of the training graph and one synthesise() call.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.make_golden as MG                    # noqa: E402  (installs the stubs, imports the reference)
from oracle import schema as S                    # noqa: E402

NSPK, NLANG, SEED = 3, 2, 5678


def main():
    c = S.SMALL
    torch.manual_seed(0)
    gen = MG.build_generator(c)
    # the reference constructor adds the embeddings when the counts are > 1 (generator/__init__.py:62-65)
    gen.num_speakers, gen.num_languages = NSPK, NLANG
    gen.sid_embed = torch.nn.Embedding(NSPK, c.dim)
    gen.lid_embed = torch.nn.Embedding(NLANG, c.dim)
    gen.train()
    schema = S.generator_schema(c)
    schema["generator.sid_embed.weight"] = (NSPK, c.dim)
    schema["generator.lid_embed.weight"] = (NLANG, c.dim)
    weights = S.make_weights(schema, SEED)
    MG.load_weights(gen, weights, "generator.")
    B = 3
    batch = MG.make_batch(c, B, (17, 24), (90, 120), SEED + 1, wav=False)
    tb = MG.to_t(batch)
    sids = torch.tensor([2, 0, 1])
    lids = torch.tensor([1, 1, 0])
    rand01 = np.random.default_rng(SEED + 2).uniform(0, 1, B).astype(np.float32)

    def _grs(x, x_lengths, segment_size):
        max_start = x_lengths - segment_size
        max_start[max_start < 0] = 0
        starts = (torch.from_numpy(rand01) * max_start).to(dtype=torch.long)
        return MG.RSeg.get_segments(x, starts, segment_size), starts
    MG.RG.get_random_segments = _grs
    cap = {}
    _vd = MG.RA.viterbi_decode

    def _viterbi(lp, tl, fl):
        ds, bl = _vd(lp, tl, fl)
        cap["durations"] = ds.detach().clone()
        return ds, bl
    MG.RG.viterbi_decode = _viterbi
    out = gen(x=tb["x"], x_lengths=tb["x_lengths"], mel=tb["mel"], mel_lengths=tb["mel_lengths"], pitches=tb["pitches"],
              energies=tb["energies"], sids=sids, lids=lids)
    gen.zero_grad()
    out["loss"].backward()
    res = dict(seed=np.int64(SEED), rand01=rand01, sids=sids.numpy(), lids=lids.numpy(),
               **{"in_" + k: v for k, v in batch.items()})
    res["loss"] = out["loss"].detach().numpy()
    for k in ("align_loss", "duration_loss", "pitch_loss", "energy_loss"):
        res[k] = out[k].numpy()
    res["durations"] = cap["durations"].numpy()
    res["start_idx"] = out["start_idx"].numpy()
    res["wav_hat"] = out["wav_hat"].detach().numpy()
    res["grad_sid_embed"] = gen.sid_embed.weight.grad.numpy()
    res["grad_lid_embed"] = gen.lid_embed.weight.grad.numpy()
    names, norms = [], []
    for k, p in gen.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(p.grad.double().norm().item())
    res["grad_g_names"], res["grad_g_norms"] = np.array(names), np.array(norms)
    # inference with explicit ids, and with ids omitted (the reference then uses speaker / language 0, :235-240)
    gen.eval()
    with torch.no_grad():
        gen.duration_predictor.linear.bias.fill_(1.2)
    x_len = np.array([21, 13, 17])
    g = np.random.default_rng(SEED + 5)
    x = g.integers(1, 159, (3, 21))
    for b in range(3):
        x[b, x_len[b]:] = 0
    for tag, kw in (("syn", dict(sids=sids, lids=lids)), ("syn0", dict())):
        o = gen.synthesise(torch.from_numpy(x), torch.from_numpy(x_len), d_factor=1.1, p_factor=1.6, e_factor=1.2, **kw)
        res[tag + "_wav"], res[tag + "_durations"] = o["wav"].numpy(), o["durations"].numpy()
        res[tag + "_wav_lengths"] = o["wav_lengths"].numpy()
    res["syn_x"], res["syn_x_lengths"], res["dur_bias"] = x.astype(np.int64), x_len.astype(np.int64), np.float32(1.2)
    np.savez_compressed(os.path.join(MG.OUT, "gen_small_multispk.npz"), **res)
    print("gen_small_multispk loss", float(res["loss"]), "MAS shim==pairwise:", all(MG.MAS_AGREE), len(MG.MAS_AGREE),
          "dur sums", res["syn_durations"].sum(1), res["syn0_durations"].sum(1))


if __name__ == "__main__":
    main()
