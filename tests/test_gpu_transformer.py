"""A19: Transformer backbone variant on the HIP kernels vs the reference golden / the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_transformer_vs_reference_golden(golden):
    """forward + every gradient of the 2-block reference run (ragged batch, eval mode), f32 mode, 1e-3 of the tensor scale"""
    from optispeech_amd import precision
    from optispeech_amd.model.transformer import Transformer
    precision.set_precision("f32")
    g = golden("transformer")
    m = Transformer(dim=64, linear_units=96, num_blocks=2).to(DEV).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(g["w_" + k]) for k in g["keys"].tolist()}, strict=True)
    assert not missing and not unexpected
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    lens = torch.from_numpy(g["lens"]).to(DEV)
    pad = torch.arange(x.shape[1], device=DEV)[None] >= lens[:, None]
    y = m(x, pad)
    assert relerr(y, torch.from_numpy(g["y"])) < 1e-3
    (y * torch.from_numpy(g["G"]).to(DEV)).sum().backward()
    assert relerr(x.grad, torch.from_numpy(g["dx"])) < 1e-3
    sd_grads = {}
    for mprefix, mod in m.named_modules():                       # gradients in the reference layout
        for name, prm in mod._parameters.items():
            if prm is None:
                continue
            key, _, to_ref = mod._ref(name) if hasattr(mod, "_ref") else (name, None, None)
            sd_grads[(mprefix + "." if mprefix else "") + key] = to_ref(prm.grad) if to_ref else prm.grad
    for k in g["keys"].tolist():
        want = torch.from_numpy(g["g_" + k])
        # linear_k.bias has a mathematically zero gradient (a constant added to every key's score cancels in the softmax):
        # both sides hold ~1e-7 round-off there, hence the absolute floor
        err = (sd_grads[k].detach().cpu().double() - want.double()).abs().max().item()
        assert err <= 2e-3 * want.abs().max().item() + 5e-6, (k, err, want.abs().max().item())


@pytest.mark.parametrize("mode,tol", [("f32", 1e-3), ("bf16", 3e-2)])
def test_transformer_decoder_size_vs_oracle(mode, tol):
    """BASELINE config 4 decoder shape (dim 256, 2 heads, 4 blocks, T = 800), ragged batch of 4, vs the CPU oracle"""
    from optispeech_amd import precision
    from optispeech_amd.model.transformer import Transformer
    from oracle import transformer as OT
    precision.set_precision(mode)
    try:
        torch.manual_seed(0)
        m = Transformer(dim=256).to(DEV).eval()
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() == 1 and p.numel() > 1:
                    p.add_(torch.randn_like(p) * 0.05)
        B, T = 4, 800
        lens = torch.tensor([800, 643, 311, 17])
        x = torch.randn(B, T, 256, generator=torch.Generator().manual_seed(1))
        pad = torch.arange(T)[None] >= lens[:, None]
        y = m(x.to(DEV), pad.to(DEV))
        P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        want = OT.forward(P, x, pad, heads=2)
        valid = (~pad)[:, :, None]
        assert relerr(y.cpu() * valid, want * valid) < tol
    finally:
        precision.set_precision("f32")


def test_transformer_generator_train_step_runs():
    """the Transformer variant wired into the full model: one training step, finite losses, gradients reach the attention"""
    from optispeech_amd import precision, rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    precision.set_precision("bf16")
    try:
        torch.manual_seed(2)
        rng.manual_seed(2, 0)
        cfg = ModelConfig(backbone="transformer")
        m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to(DEV).train()
        batch = synthetic_batch(2, 24, 96, cfg, seed=5, device=DEV)
        og, od = m.optimizers()
        m.training_step(batch, 0)
        logs = m.fetch_logs()
        assert all(np.isfinite(v) for v in logs.values()), logs
        names = [k for k, _ in m.generator.named_parameters()]
        assert any("encoder.transformer.encoders.0.self_attn.linear_q" in k for k in names)
        gq = m.generator.encoder.transformer.encoders[0].self_attn.linear_q.weight.grad
        assert gq is not None and torch.isfinite(gq).all() and gq.abs().sum().item() > 0
        # inference path of the same variant
        from optispeech_amd.values import InferenceInputs
        x = torch.randint(1, 150, (2, 16))
        xl = torch.tensor([16, 9])
        out = m.eval().synthesise(InferenceInputs(clean_text="", x=x * (torch.arange(16)[None] < xl[:, None]), x_lengths=xl,
                                                  d_factor=1.0, p_factor=1.0, e_factor=1.0),
                                  durations_override=torch.full((2, 16), 3))
        wav = torch.as_tensor(out.wav)
        assert wav.shape[0] == 2 and torch.isfinite(wav).all()
    finally:
        precision.set_precision("f32")


@pytest.mark.parametrize("B,T,H,dk", [(3, 37, 2, 32), (2, 130, 2, 64), (4, 800, 2, 128), (1, 5, 4, 32), (2, 257, 2, 128)])
def test_fused_attention_forward_matches_the_unfused_path(B, T, H, dk):
    """osp_attn_fused_fwd (flash-style: no (T x T) scores in HBM; bf16 operands) against softmax(q k^T / sqrt(dk)) v in f64 with
    the key-padding mask of the reference (masked keys -> probability 0), ragged lengths, query blocks and key tiles that end inside
    the sequence."""
    from optispeech_amd import kernels as K
    g = torch.Generator().manual_seed(B * 100 + T)
    C = H * dk
    q, k, v = (torch.randn(B, T, C, generator=g).to(DEV) for _ in range(3))
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    klen = lens.to(DEV)
    o = K.attn_fused_fwd(q, k, v, klen, H)
    bf = lambda t: t.to(torch.bfloat16).double().cpu()                                            # noqa: E731
    qh, kh, vh = (bf(t).view(B, T, H, dk).permute(0, 2, 1, 3) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / dk ** 0.5
    mask = torch.arange(T)[None, None, None, :] >= lens[:, None, None, None]
    p = torch.softmax(s.masked_fill(mask, float("-inf")), -1).masked_fill(mask, 0.0)
    want = (p @ vh).permute(0, 2, 1, 3).reshape(B, T, C)
    err = (o.double().cpu() - want).abs().max().item()
    assert err < 2e-2, err                                                                         # probabilities are rounded to bf16 for the P V product
    rel = ((o.double().cpu() - want).norm() / want.norm()).item()
    assert rel < 5e-3, rel


def test_attention_kernels_clamp_lengths_beyond_T():
    """the Transformer backbone hands the RAW lengths tensor to the attention kernels (model/transformer.py), and a
    raw length may exceed the T of the tensor (an inconsistent batch, a mask built for a longer T and sliced) where a mask's row sum
    could not.  Every kernel clamps -- kl = min(T, klen[b]) -- so lengths beyond T behave exactly as lengths == T: bit-identical
    results, nothing read or written past the T keys (a canary row behind every operand stays untouched)."""
    from optispeech_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    B, T, H, dk = 2, 70, 2, 64
    C, Z = H * dk, B * H
    q, k, v, dout = (torch.randn(B, T, C, generator=g).to(DEV) for _ in range(4))
    at_T = torch.tensor([T, 33], device=DEV)
    beyond = torch.tensor([T + 900, 33], device=DEV)
    assert torch.equal(K.attn_fused_fwd(q, k, v, at_T, H), K.attn_fused_fwd(q, k, v, beyond, H))
    o0, l0 = K.attn_train_fwd(q, k, v, at_T, H, 0.2, 99, 3)
    o1, l1 = K.attn_train_fwd(q, k, v, beyond, H, 0.2, 99, 3)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    for a, b in zip(K.attn_train_bwd(q, k, v, o0, l0, dout, at_T, H, 0.2, 99, 3), K.attn_train_bwd(q, k, v, o0, l0, dout, beyond, H, 0.2, 99, 3)):
        assert torch.equal(a, b)
    S = torch.randn(Z + 1, T, T, generator=g).to(DEV)                  # one canary (T x T) block behind the scores
    canary = S[Z].clone()
    S0, S1 = S.clone(), S.clone()
    P0, Pd0 = K.attn_softmax_fwd(S0[:Z], at_T, B, H, T, T, 0.125, 0.2, 5, 1)
    P1, Pd1 = K.attn_softmax_fwd(S1[:Z], beyond, B, H, T, T, 0.125, 0.2, 5, 1)
    assert torch.equal(P0, P1) and torch.equal(Pd0, Pd1)
    assert torch.equal(S1[Z], canary)


def test_attention_function_takes_the_fused_kernel_without_grad():
    from optispeech_amd import ops, precision
    precision.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(3)
        q, k, v = (torch.randn(2, 96, 256, generator=g).to(DEV) for _ in range(3))
        klen = torch.tensor([96, 40], device=DEV)
        with torch.no_grad():
            fused = ops.AttentionFn.apply(q, k, v, klen, 2, 0.0, 0, 0)
        keep = ops._FUSED_ATTN_TRAIN
        ops._FUSED_ATTN_TRAIN = False
        try:
            ref = ops.AttentionFn.apply(q.requires_grad_(True), k, v, klen, 2, 0.0, 0, 0)              # unfused (saves probabilities)
        finally:
            ops._FUSED_ATTN_TRAIN = keep
        assert ref.grad_fn is not None
        torch.testing.assert_close(fused, ref.detach(), rtol=3e-2, atol=3e-2)
    finally:
        precision.set_precision("f32")


def _dropout_keep(Z, T, p, seed, stream_id):
    """The keep / (1 - p) factors the attention kernels draw for element (z, i, j): read back from the UNFUSED softmax kernel on
    all-zero scores (P is uniform and positive there, so Pd / P is the factor)."""
    from optispeech_amd import kernels as K
    if p == 0.0:
        return torch.ones(Z, T, T, dtype=torch.float64)
    S = torch.zeros(Z, T, T, device=DEV)
    P, Pd = K.attn_softmax_fwd(S, torch.full((Z,), T, device=DEV, dtype=torch.int64), Z, 1, T, T, 1.0, p, seed, stream_id)
    return (Pd / P).double().cpu()


@pytest.mark.parametrize("B,T,H,dk,p", [(3, 37, 2, 32, 0.0), (2, 130, 2, 64, 0.2), (2, 800, 2, 128, 0.2), (1, 5, 4, 32, 0.3),
                                        (2, 257, 2, 128, 0.0), (2, 131, 2, 128, 0.2)])
def test_fused_attention_training_pair_vs_f64(B, T, H, dk, p):
    """osp_attn_train_fwd / osp_attn_train_bwd (csrc/attention_train.hip: no (T x T) tensor in HBM, probabilities recomputed in the
    backward) against MultiHeadedAttention's arithmetic (_transformer/attention.py:80-98, :120-125: masked softmax -> masked_fill 0
    -> dropout -> P V) in f64 autograd, on bf16-rounded operands and with the SAME Philox dropout mask.  Ragged key lengths,
    query / key blocks that end inside the sequence, T not a multiple of 4 (the Philox block of an element then straddles rows).
    Tolerances: bf16 rounding of P / dS before their second product (unit round-off 4e-3): 1e-2 of the tensor's norm."""
    from optispeech_amd import kernels as K
    g = torch.Generator().manual_seed(B * 1000 + T)
    C, Z = H * dk, B * H
    q, k, v, dout = (torch.randn(B, T, C, generator=g).to(DEV) for _ in range(4))
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    klen = lens.to(DEV)
    seed, sid = 4242, 7
    o, lse = K.attn_train_fwd(q, k, v, klen, H, p, seed, sid)
    dq, dk_, dv = K.attn_train_bwd(q, k, v, o, lse, dout, klen, H, p, seed, sid)
    keep = _dropout_keep(Z, T, p, seed, sid).view(B, H, T, T)
    bf = lambda t: t.to(torch.bfloat16).double().cpu()                                            # noqa: E731
    q64, k64, v64 = (bf(t).requires_grad_(True) for t in (q, k, v))
    heads = lambda t: t.view(B, T, H, dk).permute(0, 2, 1, 3)                                     # noqa: E731
    s = heads(q64) @ heads(k64).transpose(-1, -2) / dk ** 0.5
    mask = torch.arange(T)[None, None, None, :] >= lens[:, None, None, None]
    P = torch.softmax(s.masked_fill(mask, float("-inf")), -1).masked_fill(mask, 0.0)
    want = ((P * keep) @ heads(v64)).permute(0, 2, 1, 3).reshape(B, T, C)
    want.backward(bf(dout))
    rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()        # noqa: E731
    assert rel(o, want.detach()) < 1e-2
    want_lse = torch.logsumexp(s.detach().masked_fill(mask, float("-inf")), -1).reshape(Z, T)
    assert (lse.double().cpu() - want_lse).abs().max().item() < 2e-2
    assert rel(dq, q64.grad) < 1.5e-2 and rel(dk_, k64.grad) < 1.5e-2 and rel(dv, v64.grad) < 1.5e-2
    # keys past an utterance's length receive exactly zero gradient
    for b in range(B):
        assert torch.count_nonzero(dk_[b, int(lens[b]):]) == 0 and torch.count_nonzero(dv[b, int(lens[b]):]) == 0


def test_attention_function_trains_through_the_fused_pair():
    """In the performance mode AttentionFn routes a training call through the fused pair; same dropout mask as the unfused path,
    so the two agree to the bf16 rounding of their operands (outputs and all three gradients)."""
    from optispeech_amd import ops, precision
    precision.set_precision("bf16")
    keep = ops._FUSED_ATTN_TRAIN
    try:
        g = torch.Generator().manual_seed(11)
        base = [torch.randn(2, 200, 256, generator=g).to(DEV) for _ in range(3)]
        dout = torch.randn(2, 200, 256, generator=g).to(DEV)
        klen = torch.tensor([200, 77], device=DEV)
        res = {}
        for fused in (True, False):
            ops._FUSED_ATTN_TRAIN = fused
            q, k, v = (t.clone().requires_grad_(True) for t in base)
            o = ops.AttentionFn.apply(q, k, v, klen, 2, 0.2, 99, 3)
            o.backward(dout)
            res[fused] = (o.detach(), q.grad, k.grad, v.grad)
        for a, b in zip(res[True], res[False]):
            err = ((a - b).norm() / b.norm()).item()
            assert err < 2e-2, err
    finally:
        ops._FUSED_ATTN_TRAIN = keep
        precision.set_precision("f32")


@pytest.mark.parametrize("B,T,H,dk,p", [(2, 130, 2, 64, 0.2), (2, 257, 4, 32, 0.0), (1, 800, 2, 128, 0.1)])
def test_fused_attention_with_a_score_term_vs_f64(B, T, H, dk, p):
    """The relative-position form (RelPositionMultiHeadedAttention, _transformer/attention.py:290-313: scores = (q k^T + bd) / sqrt(dk)):
    the fused pair with an additive (B*H, T, T) score term against f64 autograd -- output, dq / dk / dv and the term's own gradient
    (= dS, written by the dQ kernel in f32), ragged key lengths (the gradient is exactly zero on masked keys)."""
    from optispeech_amd import kernels as K
    g = torch.Generator().manual_seed(B * 1000 + T + 5)
    C, Z = H * dk, B * H
    q, k, v, dout = (torch.randn(B, T, C, generator=g).to(DEV) for _ in range(4))
    sb = (torch.randn(Z, T, T, generator=g) * 2.0).to(DEV)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    klen = lens.to(DEV)
    seed, sid = 77, 5
    o, lse = K.attn_train_fwd(q, k, v, klen, H, p, seed, sid, sbias=sb)
    dq, dk_, dv, dsb = K.attn_train_bwd(q, k, v, o, lse, dout, klen, H, p, seed, sid, sbias=sb, want_dsbias=True)
    keep = _dropout_keep(Z, T, p, seed, sid).view(B, H, T, T)
    bf = lambda t: t.to(torch.bfloat16).double().cpu()                                            # noqa: E731
    q64, k64, v64 = (bf(t).requires_grad_(True) for t in (q, k, v))
    sb64 = sb.double().cpu().view(B, H, T, T).requires_grad_(True)
    heads = lambda t: t.view(B, T, H, dk).permute(0, 2, 1, 3)                                     # noqa: E731
    s = (heads(q64) @ heads(k64).transpose(-1, -2) + sb64) / dk ** 0.5
    mask = torch.arange(T)[None, None, None, :] >= lens[:, None, None, None]
    P = torch.softmax(s.masked_fill(mask, float("-inf")), -1).masked_fill(mask, 0.0)
    want = ((P * keep) @ heads(v64)).permute(0, 2, 1, 3).reshape(B, T, C)
    want.backward(bf(dout))
    rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()        # noqa: E731
    assert rel(o, want.detach()) < 1e-2
    assert rel(dq, q64.grad) < 1.5e-2 and rel(dk_, k64.grad) < 1.5e-2 and rel(dv, v64.grad) < 1.5e-2
    assert rel(dsb.view(B, H, T, T), sb64.grad) < 1.5e-2
    for b in range(B):
        assert torch.count_nonzero(dsb.view(B, H, T, T)[b, :, :, int(lens[b]):]) == 0


def test_attention_function_with_score_term_fused_vs_unfused():
    """AttentionFn with sbias (the Conformer's call): fused pair vs the unfused kernels in the performance mode, same dropout mask."""
    from optispeech_amd import ops, precision
    precision.set_precision("bf16")
    keep = ops._FUSED_ATTN_TRAIN
    try:
        g = torch.Generator().manual_seed(12)
        base = [torch.randn(2, 150, 256, generator=g).to(DEV) for _ in range(3)]
        sb0 = torch.randn(2 * 4, 150, 150, generator=g).to(DEV)
        dout = torch.randn(2, 150, 256, generator=g).to(DEV)
        klen = torch.tensor([150, 61], device=DEV)
        res = {}
        for fused in (True, False):
            ops._FUSED_ATTN_TRAIN = fused
            q, k, v = (t.clone().requires_grad_(True) for t in base)
            sb = sb0.clone().requires_grad_(True)
            o = ops.AttentionFn.apply(q, k, v, klen, 4, 0.1, 99, 3, sb)
            o.backward(dout)
            res[fused] = (o.detach(), q.grad, k.grad, v.grad, sb.grad)
        for a, b in zip(res[True], res[False]):
            err = ((a - b).norm() / b.norm()).item()
            assert err < 2e-2, err
    finally:
        ops._FUSED_ATTN_TRAIN = keep
        precision.set_precision("f32")


@pytest.mark.parametrize("B,T,C", [(3, 37, 64), (2, 130, 256), (1, 5, 6)])
def test_scaled_posenc_forward_backward_vs_torch(B, T, C):
    """ops.ScaledPosEncFn (osp_posenc_fwd / osp_posenc_dalpha) vs `x + alpha * pe` and its autograd (embedding.py:120-124); C = 6
    takes the scalar path of both kernels.  dx is dy itself; d alpha to f32 summation noise, twice the same bits."""
    from optispeech_amd import ops
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + T)
    x = torch.randn(B, T, C, device=DEV, generator=g)
    pe = torch.randn(T, C, device=DEV, generator=g)
    G = torch.randn(B, T, C, device=DEV, generator=g)
    outs = []
    for fn in (lambda x_, a_: ops.ScaledPosEncFn.apply(x_, a_, pe), lambda x_, a_: x_ + a_ * pe):
        xx = x.clone().requires_grad_(True)
        al = torch.tensor(0.7, device=DEV, requires_grad=True)
        y = fn(xx, al)
        (y * G).sum().backward()
        outs.append((y.detach(), xx.grad, al.grad))
    (y, dx, da), (yr, dxr, dar) = outs
    assert da.shape == dar.shape
    assert (y - yr).abs().max().item() <= 1e-6 * max(1.0, yr.abs().max().item())
    assert torch.equal(dx, dxr)
    assert abs(da.item() - dar.item()) <= 1e-4 * max(1.0, abs(dar.item()))
    xx = x.clone().requires_grad_(True)
    al = torch.tensor(0.7, device=DEV, requires_grad=True)
    (ops.ScaledPosEncFn.apply(xx, al, pe) * G).sum().backward()
    assert torch.equal(al.grad, da)                              # fixed summation order: bit-reproducible


@pytest.mark.parametrize("shape", [(2, 7, 3, 16), (32, 128, 2, 128), (1, 1, 5, 4)])
def test_permute_0213_vs_torch(shape):
    """osp_permute_0213 == x.transpose(1, 2).contiguous() (the head split / merge of attention.py:59-66,99-101), both directions"""
    from optispeech_amd import kernels as K
    x = torch.randn(*shape, device=DEV)
    y = K.permute_0213(x)
    assert torch.equal(y, x.transpose(1, 2).contiguous())
    assert torch.equal(K.permute_0213(y), x)


def test_transformer_dropout_sites_use_the_documented_philox_counters():
    """_dropout (osp_dropout_add) == x * keep-mask of the same (seed, stream) + residual, bit for bit: the mask a LayerNorm launch
    with drop_p materialises (ops.dropout_mask) uses counter = element index / 4 as well."""
    from optispeech_amd import ops, rng
    from optispeech_amd.model.transformer import _dropout
    rng.manual_seed(11, 0)
    x = torch.randn(4, 50, 64, device=DEV)
    res = torch.randn(4, 50, 64, device=DEV)
    y = _dropout(x, 0.2, True, 5, res=res)
    mask = ops.dropout_mask((200, 64), 0.2, rng.seed(), 5, DEV).view(4, 50, 64)
    assert torch.equal(y, res + x * mask)
    assert 0.1 < (mask == 0).float().mean().item() < 0.3
