"""The split-bf16 conv-GEMM (csrc/gemm_f32_split.hip behind osp_conv_gemm_f32_split): f32 operands in HBM, every operand element
entering the matrix pipe as a (hi, lo) pair of bf16 numbers, three bf16 MFMAs per product, f32 accumulate -- the kernel the "mixed"
parity mode runs the generator's GEMMs on outside the index-critical path (optispeech_amd/precision.py).

Tolerance, stated: what the split drops is <= 3 x 2^-18 = 1.1e-5 of |a b| PER PRODUCT; over a K-long dot product with operands of
either sign that is ~4e-6 / sqrt(K)-ish of sum |a||b|.  The tests bound max |got - f64| by 1.2e-5 x the row's sum_k |a_k||b_k|
(a bound that holds for ANY signs) and, against the result's own scale, by 3e-5 (the exact kernel holds 2e-6 on the same shapes,
plain bf16 operands ~4e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _symbol():
    from optispeech_amd import _lib
    import ctypes
    buf = ctypes.create_string_buffer(128)
    fl = ctypes.c_double(0)
    _lib.lib().cdll.osp_kernel_note_host(buf, 128, ctypes.byref(fl))
    return buf.value.decode()


@pytest.fixture
def mixed():
    from optispeech_amd import precision
    precision.set_precision("mixed")
    try:
        yield precision
    finally:
        precision.set_precision("f32")


# (utterances, frames, Cin, taps, Cout): 128 x 128 tiles, 128 x 64 tiles, 64 x 64 tiles, ragged row counts, Cin = 96 (three 32-deep slabs)
SHAPES = [(64, 128, 256, 1, 1024), (40, 413, 256, 5, 256), (33, 391, 384, 3, 384), (7, 1999, 64, 7, 1152), (64, 128, 96, 1, 512),
          (64, 128, 1024, 1, 256), (32, 64, 384, 1, 1152), (32, 64, 1152, 1, 384), (16, 128, 256, 3, 256)]


@pytest.mark.parametrize("nutt,T,cin,taps,n_out", SHAPES)
def test_split_gemm_forward_convs_vs_f64(nutt, T, cin, taps, n_out, mixed):
    from optispeech_amd import kernels as K
    assert mixed.f32_split()
    pad = (taps - 1) // 2
    x = rnd(nutt, T, cin, seed=1)
    w = rnd(n_out, taps, cin, seed=2, scale=1.0 / np.sqrt(cin * taps))
    b = rnd(n_out, seed=3)
    xt, wt = x.double().transpose(1, 2), w.double().permute(0, 2, 1)
    want = F.conv1d(xt, wt, b.double(), padding=pad).transpose(1, 2)
    mag = F.conv1d(xt.abs(), wt.abs(), None, padding=pad).transpose(1, 2)          # sum_k |a_k| |b_k| per output element
    got = K.conv_gemm(x.to(DEV).view(nutt * T, cin), w.to(DEV), n_out, T=T, taps=taps, pad=pad, bias=b.to(DEV))
    assert _symbol() == "conv_gemm_f32_split_kernel"
    err = (got.view(nutt, T, n_out).cpu().double() - want).abs()
    assert (err <= 1.2e-5 * mag + 1e-6 * want.abs()).all(), (err / mag).max().item()
    assert relerr(got.view(nutt, T, n_out), want) < 3e-5
    got_r = K.conv_gemm(x.to(DEV).view(nutt * T, cin), w.to(DEV), n_out, T=T, taps=taps, pad=pad, bias=b.to(DEV), epi=K.EPI_RELU)
    assert relerr(got_r.view(nutt, T, n_out), F.relu(want)) < 3e-5


def test_split_gemm_block_epilogues(mixed):
    from optispeech_amd import kernels as K
    M, C, I = 8191, 256, 1024                                        # ragged: the last row tile is one row short
    h, W1, b1 = rnd(M, C, seed=1), rnd(I, C, seed=2, scale=0.06), rnd(I, seed=3, scale=0.1)
    u_want = F.linear(h.double(), W1.double(), b1.double())
    u = torch.empty(M, I, device=DEV)
    g = K.conv_gemm(h.to(DEV), W1.to(DEV), I, epi=K.EPI_GELU, bias=b1.to(DEV), aux_out=u)
    assert _symbol() == "conv_gemm_f32_split_kernel"
    assert relerr(u, u_want) < 3e-5 and relerr(g, F.gelu(u_want)) < 3e-5
    W2, b2, gam = rnd(C, I, seed=4, scale=0.03), rnd(C, seed=5, scale=0.1), rnd(C, seed=6)
    res, mask, rs = rnd(M, C, seed=7), (torch.arange(M) % 7 != 0).float(), torch.rand(M, generator=torch.Generator().manual_seed(8))
    gd = g.cpu().double()
    z_want = F.linear(gd, W2.double(), b2.double())
    y_want = (res.double() + rs.double()[:, None] * gam.double() * z_want) * mask.double()[:, None]
    z = torch.empty(M, C, device=DEV)
    y = K.conv_gemm(g, W2.to(DEV), C, epi=K.EPI_SCALE_RES_MASK, bias=b2.to(DEV), gamma=gam.to(DEV), res=res.to(DEV),
                    rowmask=mask.to(DEV), rowscale=rs.to(DEV), aux_out=z)
    assert _symbol() == "conv_gemm_f32_split_kernel"
    assert relerr(z, z_want) < 3e-5 and relerr(y, y_want) < 3e-5
    base = rnd(M, C, seed=11).to(DEV)
    acc = base.clone()
    K.conv_gemm(g, W2.to(DEV), C, out=acc, accumulate=True)
    assert relerr(acc, base.cpu().double() + F.linear(gd, W2.double())) < 3e-5


def test_split_is_off_on_the_index_path_and_in_the_other_modes(mixed):
    """Inside precision.index_path() -- and in the f32 / bf16 modes -- the same call takes the exact kernel: durations and alignment
    indices come from the kernels the f32 mode runs, bit for bit."""
    from optispeech_amd import kernels as K, precision
    x, w, b = rnd(64, 128, 256, seed=1).to(DEV), rnd(1024, 3, 256, seed=2, scale=0.04).to(DEV), rnd(1024, seed=3).to(DEV)
    with precision.index_path():
        assert not precision.f32_split()
        inside = K.conv_gemm(x.view(-1, 256), w, 1024, T=128, taps=3, pad=1, bias=b)
        assert _symbol() == "conv_gemm_f32_glds_kernel"
        precision.leave_index_path()
        assert precision.f32_split()
    assert precision.f32_split()
    split = K.conv_gemm(x.view(-1, 256), w, 1024, T=128, taps=3, pad=1, bias=b)
    assert _symbol() == "conv_gemm_f32_split_kernel"
    precision.set_precision("f32")
    assert not precision.f32_split()
    exact = K.conv_gemm(x.view(-1, 256), w, 1024, T=128, taps=3, pad=1, bias=b)
    assert _symbol() == "conv_gemm_f32_glds_kernel"
    assert torch.equal(inside, exact)
    assert 0 < relerr(split, exact) < 3e-5
    precision.set_precision("bf16")
    assert not precision.f32_split()
    with precision.index_path():
        assert not precision.f32_split()
    precision.set_precision("mixed")
    precision.set_f32_split(False)
    try:
        assert not precision.f32_split()
    finally:
        precision.set_f32_split(True)


def test_split_gemm_rows_independent_of_the_batch(mixed):
    from optispeech_amd import kernels as K
    x, w, b = rnd(64, 128, 256, seed=1).to(DEV), rnd(1024, 3, 256, seed=2, scale=0.04).to(DEV), rnd(1024, seed=3).to(DEV)
    full = K.conv_gemm(x.view(-1, 256), w, 1024, T=128, taps=3, pad=1, bias=b).view(64, 128, 1024)
    part = K.conv_gemm(x[16:48].reshape(-1, 256), w, 1024, T=128, taps=3, pad=1, bias=b).view(32, 128, 1024)
    assert torch.equal(full[16:48], part)


@pytest.mark.parametrize("M,N,C,taps,T,arow,batch", [
    (2048, 384, 1152, 1, 2048, False, 1),          # vocoder pointwise pair
    (4096, 256, 256, 5, 128, True, 1),             # variance-predictor conv with the padding mask as row factor
    (1024, 128, 192, 3, 128, True, 2),             # a batch of two problems
    (1600, 256, 100, 3, 800, False, 1),            # alignment feature conv: Cin = 100
    (25600, 256, 1024, 1, 800, True, 1),           # decoder pointwise conv at the BASELINE shape (400 slabs per tile)
])
def test_split_wgrad_ring_vs_f64(M, N, C, taps, T, arow, batch, mixed):
    """osp_conv_wgrad_f32_split_ws (conv_wgrad_ring_f32_kernel<., SPLIT>) against an f64 restatement: the split drops <= 1.1e-5 of
    |dy x| per product; bound 3e-5 of the gradient's scale (the exact kernel holds 2e-5, dominated by f32 summation).  The bias
    gradient is an exact f32 sum.  Two runs agree bit for bit (no atomics)."""
    from optispeech_amd import kernels as K
    g0 = torch.Generator().manual_seed(11)
    x, dy = torch.randn(batch, M, C, generator=g0) * 0.5, torch.randn(batch, M, N, generator=g0) * 0.5
    osc = torch.rand(N, generator=torch.Generator().manual_seed(13)) + 0.5
    ar = (torch.rand(batch, M, generator=torch.Generator().manual_seed(14)) > 0.2).float() * 1.25 if arow else None
    pad = taps // 2
    xs = x.double().view(batch, M // T, T, C)
    ys = dy.double().view(batch, M // T, T, N)
    if ar is not None:
        ys = ys * ar.double().view(batch, M // T, T, 1)
    want_w = torch.zeros(batch, N, taps, C, dtype=torch.float64)
    for j in range(taps):
        sh = j - pad
        lo, hi = max(0, -sh), min(T, T - sh)
        want_w[:, :, j, :] = torch.einsum("butn,butc->bnc", ys[:, :, lo:hi], xs[:, :, lo + sh:hi + sh])
    want_w *= osc.double()[None, :, None, None]
    want_b = ys.sum((1, 2)) * osc.double()[None]
    g = torch.Generator().manual_seed(15)
    w0, b0 = torch.randn(batch, N, taps, C, generator=g), torch.randn(batch, N, generator=g)
    outs = []
    for _ in range(2):
        dw, db = w0.to(DEV).clone(), b0.to(DEV).clone()
        if batch == 1:
            K.conv_wgrad(dy[0].to(DEV), x[0].to(DEV), dw, db, T=T, taps=taps, pad=pad, arow=None if ar is None else ar[0].to(DEV), oscale=osc.to(DEV))
        else:
            K.conv_wgrad(dy.to(DEV), x.to(DEV), dw, db, T=T, taps=taps, pad=pad, arow=None if ar is None else ar.to(DEV), oscale=osc.to(DEV),
                         batch=batch)
        assert _symbol() == "conv_wgrad_ring_f32_split_kernel"
        outs.append((dw.cpu(), db.cpu()))
    (dw, db), (dw2, db2) = outs
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "no atomics: two runs must agree bit for bit"
    assert ((dw.double() - w0.double()) - want_w).abs().max().item() <= 3e-5 * want_w.abs().max().item() + 1e-6
    assert ((db.double() - b0.double()) - want_b).abs().max().item() <= 2e-5 * want_b.abs().max().item() + 1e-6
