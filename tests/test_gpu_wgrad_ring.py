"""Ring-pipelined, atomic-free weight gradients (csrc/wgrad_ring.hip) against an f64 autograd restatement of the convolution
whose weight gradient they are (the op: torch autograd of nn.Conv2d / nn.Conv1d at
optispeech/model/vocoder/wavenext/disc/_discriminators.py:51-60,154-163 and generator/modules/convnext.py:39-41).
Tolerance: operands are bf16 values held exactly in f64 by the restatement, products accumulate in f32: 2e-4 of the
gradient's scale (stated per case)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * 0.5).to(torch.bfloat16)


def _ref2d(x, dy, KH, KW, sh, sw, ph, pw):
    """x (U,H,W,C), dy (U,Ho,Wo,N) bf16 -> dW (N,KH,KW,C), db (N) in f64 through torch autograd on the CPU."""
    xd = x.double().permute(0, 3, 1, 2).contiguous()
    N, C = dy.shape[-1], x.shape[-1]
    w = torch.zeros(N, C, KH, KW, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(N, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xd, w, b, stride=(sh, sw), padding=(ph, pw))
    y.backward(dy.double().permute(0, 3, 1, 2).contiguous())
    return w.grad.permute(0, 2, 3, 1).contiguous(), b.grad


CASES_2D = [
    # U, H, W, C, N, KH, KW, sh, sw, ph, pw      (DiscriminatorR layer geometries at reduced size, DiscriminatorP as KH = 1)
    (3, 9, 37, 64, 64, 3, 5, 1, 2, 1, 2),
    (3, 9, 37, 64, 64, 3, 5, 2, 2, 1, 2),
    (2, 7, 21, 64, 64, 3, 3, 1, 2, 1, 1),
    (2, 7, 21, 64, 64, 3, 3, 2, 2, 1, 1),
    (5, 1, 301, 128, 256, 1, 5, 1, 3, 0, 2),       # 128-tiles, stride 3
    (5, 1, 301, 32, 128, 1, 5, 1, 3, 0, 2),        # 32 input channels: upper half of the tile reads zeros
    (4, 1, 150, 128, 128, 1, 5, 1, 1, 0, 2),
]


@pytest.mark.parametrize("case", CASES_2D)
@pytest.mark.parametrize("one_split", [False, True])
def test_conv2d_wgrad_ring_vs_f64(case, one_split):
    from optispeech_amd import kernels as K
    U, H, W, C, N, KH, KW, sh, sw, ph, pw = case
    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    x, dy = _bf(U, H, W, C, seed=1), _bf(U, Ho, Wo, N, seed=2)
    want_w, want_b = _ref2d(x, dy, KH, KW, sh, sw, ph, pw)
    dev = "cuda"
    xg, dyg = x.to(dev), dy.to(dev)
    g = torch.Generator().manual_seed(3)
    w0, b0 = torch.randn(N, KH, KW, C, generator=g), torch.randn(N, generator=g)
    outs = []
    for _ in range(2):
        dw, db = w0.to(dev).clone(), b0.to(dev).clone()
        M = U * Ho * Wo
        blk = (N * KH * KW * C + N + 3) // 4 * 16
        ws = torch.empty((blk if one_split else 64 * blk,), device=dev, dtype=torch.uint8)       # one block: no room for a split
        K.call("osp_conv2d_wgrad_bf16_ws", dyg.view(M, N), 1, N, xg.view(U * H * W, C), 1, C, M, Ho * Wo, Wo, H, W, N, C, KH * KW, KW, ph,
               pw, sh, sw, dw, db, ws, ws.numel())
        outs.append((dw.cpu(), db.cpu()))
    (dw, db), (dw2, db2) = outs
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "no atomics: two runs must agree bit for bit"
    sw_, sb_ = want_w.abs().max().item(), want_b.abs().max().item()
    assert ((dw.double() - w0.double()) - want_w).abs().max().item() <= 2e-4 * sw_ + 1e-6       # `+=` into the existing gradient
    assert ((db.double() - b0.double()) - want_b).abs().max().item() <= 2e-4 * sb_ + 1e-6


@pytest.mark.parametrize("M,N,C,taps,T", [(2048, 384, 1152, 1, 2048), (4096, 1024, 256, 1, 4096), (1024, 128, 192, 3, 128),
                                          (640, 64, 64, 1, 640)])
def test_conv_wgrad_ring_pointwise_oscale_batch(M, N, C, taps, T):
    """The generator's pointwise / k-tap weight gradients: output scale (layer-scale gamma), bias, a batch of 2 problems."""
    from optispeech_amd import kernels as K
    dev = "cuda"
    batch = 2
    x, dy = _bf(batch, M, C, seed=4), _bf(batch, M, N, seed=5)
    osc = torch.rand(N, generator=torch.Generator().manual_seed(6)) + 0.5
    pad = taps // 2
    want_w = torch.zeros(batch, N, taps, C, dtype=torch.float64)
    xs = x.double().view(batch, M // T, T, C)
    ys = dy.double().view(batch, M // T, T, N)
    for j in range(taps):
        sh = j - pad
        lo, hi = max(0, -sh), min(T, T - sh)
        want_w[:, :, j, :] = torch.einsum("butn,butc->bnc", ys[:, :, lo:hi], xs[:, :, lo + sh:hi + sh])
    want_w *= osc.double()[None, :, None, None]
    want_b = dy.double().sum(1) * osc.double()[None]
    dw = torch.zeros(batch, N, taps, C, device=dev)
    db = torch.zeros(batch, N, device=dev)
    K.conv_wgrad_bf16(dy.to(dev), x.to(dev), dw, db, M=M, Trows=T, Tin=T, n=N, cin=C, taps=taps, pad=pad, oscale=osc.to(dev), batch=batch,
                      strides=(M * N, M * C, N * taps * C, N))
    assert (dw.cpu().double() - want_w).abs().max().item() <= 2e-4 * want_w.abs().max().item()
    assert (db.cpu().double() - want_b).abs().max().item() <= 2e-4 * want_b.abs().max().item()


def _f32(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * 0.5


@pytest.mark.parametrize("M,N,C,taps,T,arow,batch", [
    (2048, 384, 1152, 1, 2048, False, 1),          # vocoder pointwise pair
    (4096, 256, 256, 5, 128, True, 1),             # variance-predictor conv with the padding mask as row factor
    (1024, 128, 192, 3, 128, True, 2),             # a batch of two problems
    (1600, 256, 100, 3, 800, False, 1),            # alignment feature conv: Cin = 100 (second channel tile reads zeros past 100)
    (640, 64, 64, 1, 640, False, 1),               # one split, one workgroup per tile: adds straight into dW
])
def test_conv_wgrad_f32_ring_vs_f64(M, N, C, taps, T, arow, batch):
    """Exact-f32 weight gradient on the ring kernel (conv_wgrad_ring_f32_kernel) against an f64 restatement; products are exact, sums
    are f32: 2e-5 of the gradient's scale.  Two runs must agree bit for bit (no atomics); the call adds into dW / db."""
    from optispeech_amd import kernels as K
    dev = "cuda"
    x, dy = _f32(batch, M, C, seed=11), _f32(batch, M, N, seed=12)
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
        dw, db = w0.to(dev).clone(), b0.to(dev).clone()
        if batch == 1:
            K.conv_wgrad(dy[0].to(dev), x[0].to(dev), dw, db, T=T, taps=taps, pad=pad, arow=None if ar is None else ar[0].to(dev), oscale=osc.to(dev))
        else:
            K.conv_wgrad(dy.to(dev), x.to(dev), dw, db, T=T, taps=taps, pad=pad, arow=None if ar is None else ar.to(dev), oscale=osc.to(dev),
                         batch=batch)
        outs.append((dw.cpu(), db.cpu()))
    (dw, db), (dw2, db2) = outs
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "no atomics: two runs must agree bit for bit"
    assert ((dw.double() - w0.double()) - want_w).abs().max().item() <= 2e-5 * want_w.abs().max().item() + 1e-6
    assert ((db.double() - b0.double()) - want_b).abs().max().item() <= 2e-5 * want_b.abs().max().item() + 1e-6


def _fused_reduce_cases(dev="cuda"):
    """(dW, db) of a few split weight gradients -- bf16 ring (64- and 128-tiles, 2-D taps, batch of 2) and exact-f32 ring -- as CPU tensors."""
    from optispeech_amd import kernels as K
    out = []
    for (M, N, C, taps, T, batch) in [(4096, 1024, 256, 1, 4096, 1), (2048, 384, 1152, 1, 2048, 2), (1024, 128, 192, 3, 128, 1), (3072, 64, 64, 5, 256, 1)]:
        x, dy = _bf(batch, M, C, seed=40 + taps), _bf(batch, M, N, seed=41 + taps)
        osc = torch.rand(N, generator=torch.Generator().manual_seed(6)) + 0.5
        g = torch.Generator().manual_seed(9)
        dw, db = torch.randn(batch, N, taps, C, generator=g).to(dev), torch.randn(batch, N, generator=g).to(dev)
        K.conv_wgrad_bf16(dy.to(dev), x.to(dev), dw, db, M=M, Trows=T, Tin=T, n=N, cin=C, taps=taps, pad=taps // 2, oscale=osc.to(dev), batch=batch,
                          strides=(M * N, M * C, N * taps * C, N) if batch > 1 else (0, 0, 0, 0))
        out += [dw.cpu(), db.cpu()]
    for (M, N, C, taps, T) in [(4096, 256, 256, 1, 4096), (2048, 64, 128, 3, 512)]:
        x, dy = _bf(M, C, seed=50).float().to(dev), _bf(M, N, seed=51).float().to(dev)
        arow = (torch.rand(M, generator=torch.Generator().manual_seed(8)) > 0.2).float().to(dev)
        dw, db = torch.zeros(N, taps, C, device=dev), torch.zeros(N, device=dev)
        ws = K.wgrad_workspace(N, taps, C, 1, dev)
        K.call("osp_conv_wgrad_f32_ws", dy, N, x, C, M, T, N, C, taps, taps // 2, arow, None, dw, taps * C, db, 1, 0, 0, 0, 0, ws, ws.numel())
        out += [dw.cpu(), db.cpu()]
    torch.cuda.synchronize()
    return out


def test_fused_split_reduction_equals_the_reduce_kernel_bitwise():
    """Round 6 (opt-in, OSP_WGRAD_FUSED_REDUCE=1: measured slower, csrc/wgrad_ring.hip): the last workgroup to deliver a partial of an
    output tile adds the tile's splits up (wgr_last_arriver_reduce) instead of a second launch (wgrad_split_reduce_kernel, the
    default).  Same partials, same index order, same `dW += oscale * sum`: the two forms must agree BIT FOR BIT, and the fused form
    with itself whichever workgroup arrives last."""
    import os
    import subprocess
    import sys
    ref = _fused_reduce_cases()                                  # this process: the default two-kernel form
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from tests.test_gpu_wgrad_ring import _fused_reduce_cases\n"
            "torch.save(_fused_reduce_cases(), sys.argv[1])\n") % root
    outs = []
    for k in range(2):
        out = "/tmp/osp_wgrad_fused_%d.pt" % k
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, OSP_WGRAD_FUSED_REDUCE="1"), timeout=900)
        outs.append(torch.load(out))
    assert all(torch.equal(x, y) for x, y in zip(*outs)), "fused reduction differs between two runs"
    assert len(ref) == len(outs[0])
    for i, (x, y) in enumerate(zip(outs[0], ref)):
        assert torch.equal(x, y), (i, (x - y).abs().max().item())
