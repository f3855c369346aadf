"""Seeded inputs of the full-size checksum fixtures (tests/golden/full_b32_*.npz, tools/make_golden_b32.py).

The fixtures hold what the REFERENCE computed at the benchmark's own size but not the inputs (the mel batch alone is 10 MB): the
inputs are regenerated here with the same numpy Generator call sequence the generating script used, and every regenerated
array is checked against the checksum the fixture carries -- a drifted regeneration fails instead of comparing different problems.
Test infrastructure only.
"""
import numpy as np


def cks(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.sqrt((a * a).sum())])


def make_batch(n_feats, hop, B, tt_rng, tm_rng, seed, wav=True):
    """The batch tools/make_golden.make_batch(cfg, B, tt_rng, tm_rng, seed) draws (same calls, same order)."""
    g = np.random.default_rng(seed)
    x_len = g.integers(tt_rng[0], tt_rng[1] + 1, B)
    m_len = g.integers(tm_rng[0], tm_rng[1] + 1, B)
    x_len[0], m_len[0] = tt_rng[1], tm_rng[1]
    Tt, Tm = int(x_len.max()), int(m_len.max())
    x = g.integers(1, 159, (B, Tt))
    for b in range(B):
        x[b, x_len[b]:] = 0
    mel = g.standard_normal((B, n_feats, Tm)).astype(np.float32)
    pit = g.standard_normal((B, Tm)).astype(np.float32)
    ene = g.standard_normal((B, Tm)).astype(np.float32)
    for b in range(B):
        mel[b, :, m_len[b]:] = 0
        pit[b, m_len[b]:] = 0
        ene[b, m_len[b]:] = 0
    out = dict(x=x.astype(np.int64), x_lengths=x_len.astype(np.int64), mel=mel, mel_lengths=m_len.astype(np.int64),
               pitches=pit, energies=ene)
    if wav:
        out["wav"] = g.uniform(-1, 1, (B, Tm * hop)).astype(np.float32)
    return out


def gan_batch(g, n_feats=100, hop=256):
    """Inputs of ``full_b32_gan`` from the fixture's ``batch_args``; checksums verified."""
    B, t0, t1, m0, m1, seed = (int(v) for v in g["batch_args"])
    batch = make_batch(n_feats, hop, B, (t0, t1), (m0, m1), seed)
    for k, v in batch.items():
        want = g["cks_in_" + k]
        got = cks(v)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-9), f"regenerated input {k} differs from the generating run: {got} vs {want}"
    assert np.array_equal(batch["x_lengths"], g["in_x_lengths"]) and np.array_equal(batch["mel_lengths"], g["in_mel_lengths"])
    return batch


def transformer_case(g, T=800, C=256):
    """(state dict as numpy, lens, x, G) of ``full_b32_transformer``: weights, lengths, input and output cotangent in the generating
    script's draw order."""
    rng = np.random.default_rng(int(g["seed"]))
    keys, shapes = g["keys"].tolist(), [tuple(int(v) for v in s.split(",")) if s else () for s in g["shapes"].tolist()]
    sd = {}
    for k, shp in zip(keys, shapes):
        w = np.asarray(rng.standard_normal(shp)).astype(np.float32) * np.float32(0.05 if len(shp) > 1 else 0.02)
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("after_norm.weight"):
            w = w + np.float32(1.0)
        if k.endswith("alpha"):
            w = w + np.float32(1.0)                      # init_alpha = 1.0
        sd[k] = np.asarray(w, dtype=np.float32)
    B = len(g["lens"])
    lens = rng.integers(600, T + 1, B)
    lens[0] = T
    assert np.array_equal(lens, g["lens"])
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    G = rng.standard_normal((B, T, C)).astype(np.float32) * (np.arange(T)[None, :] < lens[:, None])[:, :, None]
    for (k, w), want in zip(sd.items(), g["w_cks"]):
        assert np.allclose(cks(w), want, rtol=1e-6, atol=1e-6), f"regenerated weight {k} differs: {cks(w)} vs {want}"
    return sd, lens.astype(np.int64), x, G.astype(np.float32)
