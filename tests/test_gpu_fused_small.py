"""Batched small kernels (csrc/fused_small.hip) against plain torch, and the bf16 ConvNeXt block (which uses them in its
backward) against the exact-f32 block."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_l1_multi_matches_torch(dtype):
    from optispeech_amd import kernels as K
    sizes = [5, 8, 4096, 8192, 8193, 70001, 3 * 8192 + 7] + [100 + 13 * i for i in range(33)]       # > 32 items: two launches
    tg = [rnd(n, seed=i).to(dtype) for i, n in enumerate(sizes)]
    ys = [rnd(n, seed=100 + i).to(dtype) for i, n in enumerate(sizes)]
    ys[2] = ys[2].float()                                      # mixed dtypes in one list (score maps are f32)
    tg[2] = tg[2].float()
    # an unaligned pair (view starting at element 1): the scalar path
    big_a, big_b = rnd(9001, seed=7).to(dtype), rnd(9001, seed=8).to(dtype)
    tg.append(big_a[1:]); ys.append(big_b[1:])
    out = torch.zeros((), device=DEV)
    K.l1_sum_multi(tg, ys, out)
    want = sum((a.float() - b.float()).abs().mean() for a, b in zip(tg, ys))
    assert abs(out.item() - want.item()) <= 2e-5 * abs(want.item())
    g = torch.tensor([0.37], device=DEV)
    gbs = K.l1_sign_multi(tg, ys, g)
    for a, b, gb in zip(tg, ys, gbs):
        ref = torch.sign(b.float() - a.float()) * (0.37 / b.numel())
        assert gb.dtype == b.dtype and torch.allclose(gb.float(), ref.to(gb.dtype).float(), rtol=1e-6 if gb.dtype == torch.float32 else 8e-3, atol=0), (b.numel(), b.dtype)


def test_hinge_multi_matches_torch():
    from optispeech_amd import kernels as K
    xs = [rnd(n, seed=n) for n in (1, 17, 4097, 9000, 20000)]
    sgns = (-1.0, 1.0, -1.0, 1.0, -1.0)
    out = torch.zeros((), device=DEV)
    K.hinge_sum_multi(xs, sgns, out)
    want = sum(torch.clamp(1 + s * x, min=0).mean() for s, x in zip(sgns, xs))
    assert abs(out.item() - want.item()) <= 1e-5 * abs(want.item())
    g = torch.tensor([2.0], device=DEV)
    for s, x, dx in zip(sgns, xs, K.hinge_grad_multi(xs, sgns, g)):
        ref = torch.where(1 + s * x > 0, torch.full_like(x, 2.0 * s / x.numel()), torch.zeros_like(x))
        assert torch.equal(dx, ref)


def test_wnorm_multi_matches_single():
    from optispeech_amd import kernels as K
    shapes = [(32, 1, 1, 5), (128, 32, 1, 5), (64, 1, 7, 5), (64, 64, 5, 3), (1, 64, 3, 3), (1024, 512, 1, 5)] * 6   # 36 convs
    vs = [rnd(*s, seed=i) for i, s in enumerate(shapes)]
    gs = [rnd(s[0], 1, 1, 1, seed=50 + i).abs() + 0.5 for i, s in enumerate(shapes)]
    many = K.wnorm_fwd_multi([(v, g, i % 2 == 0, True) for i, (v, g) in enumerate(zip(vs, gs))])
    for i, (v, g, got) in enumerate(zip(vs, gs, many)):
        want = K.wnorm_fwd(v, g, want_f32=i % 2 == 0, want_t=True)
        for a, b in zip(got, want):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b), i
    # backward: accumulate into dv / dg, compare with the per-conv kernel
    items, refs = [], []
    for i, (v, g, pk) in enumerate(zip(vs, gs, many)):
        Cout, Cin, P, Q = v.shape
        dwn = rnd(Cout, Q, P, Cin, seed=200 + i)
        dv, dg = torch.zeros_like(v), torch.zeros_like(g)
        dv2, dg2 = torch.zeros_like(v), torch.zeros_like(g)
        K.wnorm_bwd(dwn, v, g, pk[3], dv2, dg2)
        items.append((dwn, v, g, pk[3], dv, dg)); refs.append((dv2, dg2))
    K.wnorm_bwd_multi(items)
    for (_, _, _, _, dv, dg), (dv2, dg2) in zip(items, refs):
        assert torch.allclose(dv, dv2, rtol=1e-6, atol=1e-7) and torch.allclose(dg, dg2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,C", [(4096, 256), (2048, 384), (1000, 64), (77, 1024)])
def test_colsum_prod_cast_rows_pack_kscale(M, C):
    from optispeech_amd import kernels as K
    a, b, r = rnd(M, C, seed=1), rnd(M, C, seed=2), rnd(M, seed=3).abs()
    out = torch.full((C,), 0.5, device=DEV)
    K.colsum_prod(a, b, r, out)
    want = 0.5 + (a.double() * b.double() * r.double()[:, None]).sum(0)
    assert torch.allclose(out.double(), want, rtol=2e-5, atol=2e-4)
    out2 = torch.zeros((C,), device=DEV)
    K.colsum_prod(a, b, None, out2)
    assert torch.allclose(out2.double(), (a.double() * b.double()).sum(0), rtol=2e-5, atol=2e-4)
    y = K.cast_bf16_rows(a, r)
    assert torch.equal(y, (a * r[:, None]).to(torch.bfloat16))
    assert torch.equal(K.cast_bf16_rows(a, None), a.to(torch.bfloat16))
    # transposed pack with a scale along the reduction index: out[i, 0, c] = W[c, i] * gamma[c]
    I = 96
    W, gamma = rnd(C, I, seed=4), rnd(C, seed=5)
    wp = K.pack_bf16(W, I, 1, C, (1, 0, I), kscale=gamma)
    assert torch.equal(wp.view(I, C), (W * gamma[:, None]).t().contiguous().to(torch.bfloat16))


@pytest.mark.parametrize("C,I,L", [(256, 1024, 2), (384, 1152, 2)])
def test_convnext_backbone_bf16_mode_vs_f32_mode(C, I, L):
    """The bf16 block (bf16 I-wide intermediates, gamma folded into the dgrad weight pack, fused layer-scale gradient, row factor
    folded into the bf16 copy of dy) against the exact-f32 block on the same weights, ragged mask and drop-path draws: output and
    every gradient within bf16 operand tolerance."""
    from optispeech_amd import precision, rng
    from optispeech_amd.model.modules import ConvNeXtBackbone
    B, T = 3, 160
    lens = torch.tensor([160, 97, 33], device=DEV)
    pad = torch.arange(T, device=DEV)[None] >= lens[:, None]
    x0 = rnd(B, T, C, seed=1)
    gy = rnd(B, T, C, seed=2)
    res = {}
    try:
        for mode in ("f32", "bf16"):
            precision.set_precision(mode)
            torch.manual_seed(5)
            rng.reset_streams()                                   # same Philox stream id for the backbone's DropPath in both modes
            rng.manual_seed(11, 0)
            m = ConvNeXtBackbone(C, I, L, drop_path=0.3).to(DEV).train()
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if n.endswith("bias"):
                        p.add_(torch.randn_like(p) * 0.1)
            torch.manual_seed(11)
            torch.cuda.manual_seed(11)                            # same drop-path Bernoulli draws in both modes
            x = x0.clone().requires_grad_(True)
            y = m(x, pad)
            (y * gy).sum().backward()
            torch.cuda.synchronize()
            res[mode] = (y.detach(), x.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
    finally:
        precision.set_precision("f32")
    (ya, dxa, ga), (yb, dxb, gb) = res["f32"], res["bf16"]
    rel = lambda u, v: ((u.double() - v.double()).norm() / v.double().norm().clamp_min(1e-30)).item()    # noqa: E731
    assert rel(yb, ya) < 1e-2 and rel(dxb, dxa) < 2e-2, (rel(yb, ya), rel(dxb, dxa))
    for n in ga:
        assert rel(gb[n], ga[n]) < 3e-2, (n, rel(gb[n], ga[n]))


def test_segment_starts_matches_upstream_formula():
    # utils/segments.py:29-34 with the caller's num_frames = float(len - 4) (generator/__init__.py:148)
    from optispeech_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(3, 900, (37,), generator=g).cuda()
    r = torch.rand(37, generator=g).cuda()
    for seg in (0, 16, 64, 1000):
        got = K.segment_starts(r, lens, seg)
        want = (r * ((lens - 4).to(torch.float32) - seg).clamp_(min=0)).to(torch.long)
        assert torch.equal(got, want)


def test_drop_path_rows_distribution_and_mask():
    # convnext.py:121-129: bernoulli(keep) / keep per (block, utterance), constant over the frames of an utterance
    from optispeech_amd import kernels as K
    L, B, T = 6, 64, 37
    drops = [0.0, 0.1, 0.25, 0.5, 0.0, 0.9]
    rm = (torch.rand(B * T, generator=torch.Generator().manual_seed(1)) > 0.2).float().cuda()
    sc, rf = K.drop_path_rows(drops, rm, B, T, 1234, 7, rm.device)
    sc2, rf2 = K.drop_path_rows(drops, rm, B, T, 1234, 7, rm.device)
    assert torch.equal(sc, sc2) and torch.equal(rf, rf2)                    # counter-based: same key, same draw
    sc3, _ = K.drop_path_rows(drops, rm, B, T, 1235, 7, rm.device)
    assert not torch.equal(sc, sc3)
    assert torch.equal(rf, sc * rm[None])
    s3 = sc.view(L, B, T)
    assert torch.equal(s3, s3[:, :, :1].expand(-1, -1, T))                  # one factor per utterance
    for l, p in enumerate(drops):
        vals = s3[l, :, 0]
        if p == 0.0:
            assert torch.all(vals == 1.0)
        else:
            keep = 1.0 - p
            assert torch.all((vals == 0) | ((vals * keep - 1.0).abs() < 1e-5))
    # frequencies over many utterances
    scb, none = K.drop_path_rows([0.3], None, 20000, 1, 99, 3, rm.device)
    assert none is None
    frac = (scb == 0).float().mean().item()
    assert abs(frac - 0.3) < 0.02, frac


@pytest.mark.parametrize("B,T,C", [(3, 37, 256), (2, 800, 256), (4, 8, 64), (1, 5, 200), (32, 64, 384), (3, 37, 384), (2, 9, 320)])
@pytest.mark.parametrize("with_res", [True, False])
def test_fused_ln_dwconv7_backward_equals_the_two_kernel_path(B, T, C, with_res):
    """osp_ln_dwconv7_bwd against osp_layernorm_bwd followed by osp_dwconv7_bwd (which test_gpu_kernels pins to the oracle):
    same dx and same accumulated parameter gradients, ragged run lengths included (T % 8 != 0, T < window)."""
    from optispeech_amd import kernels as K
    M = B * T
    dh, xhat = rnd(M, C, seed=1), rnd(M, C, seed=2)
    rstd = rnd(M, seed=3).abs() + 0.5
    lnw, x, dw = rnd(C, seed=4), rnd(B, T, C, seed=5), rnd(7, C, seed=6)
    dres = rnd(B, T, C, seed=7) if with_res else None
    rm = (torch.rand(M, generator=torch.Generator().manual_seed(8)) > 0.3).float().to(DEV) if with_res else None
    acc0 = [rnd(C, seed=9), rnd(C, seed=10), rnd(7, C, seed=11), rnd(C, seed=12)]      # gradients accumulate onto what is there
    a = [t.clone() for t in acc0]
    dc = K.layernorm_bwd(dh, xhat, None, rstd, lnw, a[0], a[1])
    dx_ref = K.dwconv7_bwd(dc.view(B, T, C), x, dw, dres, rm, a[2], a[3])
    b = [t.clone() for t in acc0]
    dx = K.ln_dwconv7_bwd(dh, xhat, rstd, lnw, x, dw, dres, rm, b[0], b[1], b[2], b[3])
    torch.testing.assert_close(dx, dx_ref, rtol=1e-5, atol=1e-5)
    for u, v in zip(b, a):
        torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-4 * max(1.0, float(M) ** 0.5))
    # input gradient only (frozen parameters)
    dx2 = K.ln_dwconv7_bwd(dh, xhat, rstd, lnw, x, dw, dres, rm, None, None, None, None)
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("period", [2, 3, 5, 7, 11])
@pytest.mark.parametrize("B,T", [(4, 16384), (3, 1001), (2, 37)])
def test_period_fold_matches_reflect_pad_and_view(period, B, T):
    # DiscriminatorP.forward, _discriminators.py:63-72: right reflect pad to a multiple of the period, (b, t/p, p) view
    import torch.nn.functional as F
    from optispeech_amd.disc_ops import PeriodFoldFn
    x = rnd(B, T, seed=period).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    xp = xr.unsqueeze(1)
    if T % period:
        xp = F.pad(xp, (0, period - T % period), "reflect")
    tp = xp.shape[-1]
    want = xp.view(B, tp // period, period).transpose(1, 2).reshape(B * period, 1, tp // period, 1)
    got = PeriodFoldFn.apply(x, period)
    assert torch.equal(got, want)
    g = rnd(*want.shape, seed=99)
    want.backward(g)
    got.backward(g)
    torch.testing.assert_close(x.grad, xr.grad, rtol=0, atol=1e-6)


@pytest.mark.parametrize("n", [1, 7, 4096, 32 * 64 * 256 + 3])
def test_clip_matches_torch_clamp_forward_and_backward(n):
    """osp_clip (WaveNeXtHead's clip to [-1, 1], wavenext/__init__.py:47) vs torch.clip incl. its gradient rule at the limits."""
    from optispeech_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * 1.5
    x[: min(n, 3)] = torch.tensor([1.0, -1.0, 0.0])[: min(n, 3)]             # exactly on the limits: the gradient passes (<=, >=)
    xr = x.clone().requires_grad_(True)
    yr = torch.clip(xr, min=-1.0, max=1.0)
    dy = torch.randn(n, generator=g)
    yr.backward(dy)
    xg = x.to("cuda").requires_grad_(True)
    y = ops.ClipFn.apply(xg, -1.0, 1.0)
    y.backward(dy.to("cuda"))
    assert torch.equal(y.detach().cpu(), yr.detach()) and torch.equal(xg.grad.cpu(), xr.grad)
