"""The direct-to-LDS exact-f32 conv-GEMM (csrc/gemm_f32_glds.hip: v_mfma_f32_32x32x2_f32 on LDS-DMA-staged f32 tiles) behind
osp_conv_gemm_f32, at shapes that take it (>= 96 tiles of 128 x 128, Cin % 32 == 0, k-contiguous weights), against float64 torch:
plain / k-tap convs with utterance boundaries, ragged row counts, every epilogue of the ConvNeXt block, batched operands."""
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


def _took_dma():
    from optispeech_amd import _lib
    import ctypes
    buf = ctypes.create_string_buffer(128)
    fl = ctypes.c_double(0)
    _lib.lib().cdll.osp_kernel_note_host(buf, 128, ctypes.byref(fl))
    return buf.value.decode()


@pytest.mark.parametrize("nutt,T,cin,taps,n_out", [(64, 128, 256, 1, 1024), (40, 413, 256, 5, 256),
                                                   (33, 391, 384, 3, 384), (7, 1999, 64, 7, 1152), (64, 128, 96, 1, 512), (64, 128, 1024, 1, 256), (32, 128, 384, 3, 384)])
def test_f32_dma_gemm_forward_convs(nutt, T, cin, taps, n_out):
    from optispeech_amd import kernels as K
    pad = (taps - 1) // 2
    x = rnd(nutt, T, cin, seed=1)
    w = rnd(n_out, taps, cin, seed=2, scale=1.0 / np.sqrt(cin * taps))
    b = rnd(n_out, seed=3)
    want = F.conv1d(x.double().transpose(1, 2), w.double().permute(0, 2, 1), b.double(), padding=pad).transpose(1, 2)
    got = K.conv_gemm(x.to(DEV).view(nutt * T, cin), w.to(DEV), n_out, T=T, taps=taps, pad=pad, bias=b.to(DEV))
    assert _took_dma() == "conv_gemm_f32_glds_kernel"
    assert relerr(got.view(nutt, T, n_out), want) < 2e-6
    got_r = K.conv_gemm(x.to(DEV).view(nutt * T, cin), w.to(DEV), n_out, T=T, taps=taps, pad=pad, bias=b.to(DEV), epi=K.EPI_RELU)
    assert relerr(got_r.view(nutt, T, n_out), F.relu(want)) < 2e-6


def test_f32_dma_gemm_block_epilogues():
    from optispeech_amd import kernels as K
    M, C, I = 8191, 256, 1024                                        # ragged: the last row tile is one row short
    h, W1, b1 = rnd(M, C, seed=1), rnd(I, C, seed=2, scale=0.06), rnd(I, seed=3, scale=0.1)
    u_want = F.linear(h.double(), W1.double(), b1.double())
    u = torch.empty(M, I, device=DEV)
    g = K.conv_gemm(h.to(DEV), W1.to(DEV), I, epi=K.EPI_GELU, bias=b1.to(DEV), aux_out=u)
    assert _took_dma() == "conv_gemm_f32_glds_kernel"
    assert relerr(u, u_want) < 2e-6 and relerr(g, F.gelu(u_want)) < 2e-6
    W2, b2, gam = rnd(C, I, seed=4, scale=0.03), rnd(C, seed=5, scale=0.1), rnd(C, seed=6)
    res, mask, rs = rnd(M, C, seed=7), (torch.arange(M) % 7 != 0).float(), torch.rand(M, generator=torch.Generator().manual_seed(8))
    z_want = F.linear(F.gelu(u_want), W2.double(), b2.double())
    y_want = (res.double() + rs.double()[:, None] * gam.double() * z_want) * mask.double()[:, None]
    z = torch.empty(M, C, device=DEV)
    y = K.conv_gemm(g, W2.to(DEV), C, epi=K.EPI_SCALE_RES_MASK, bias=b2.to(DEV), gamma=gam.to(DEV), res=res.to(DEV),
                    rowmask=mask.to(DEV), rowscale=rs.to(DEV), aux_out=z)
    assert _took_dma() == "conv_gemm_f32_glds_kernel"
    assert relerr(z, z_want) < 3e-6 and relerr(y, y_want) < 3e-6
    ym = K.conv_gemm(g, W2.to(DEV), C, epi=K.EPI_MASK, bias=b2.to(DEV), rowmask=mask.to(DEV))
    assert relerr(ym, z_want * mask.double()[:, None]) < 3e-6
    # accumulate form
    base = rnd(M, C, seed=11).to(DEV)
    acc = base.clone()
    K.conv_gemm(g, W2.to(DEV), C, out=acc, accumulate=True)
    assert relerr(acc, base.cpu().double() + F.linear(F.gelu(u_want), W2.double())) < 3e-6


def test_f32_dma_gemm_rows_independent_of_the_batch():
    """A sub-batch reproduces its rows bit for bit (the property the exact index path relies on)."""
    from optispeech_amd import kernels as K
    x, w, b = rnd(64, 128, 256, seed=1).to(DEV), rnd(1024, 3, 256, seed=2, scale=0.04).to(DEV), rnd(1024, seed=3).to(DEV)
    full = K.conv_gemm(x.view(-1, 256), w, 1024, T=128, taps=3, pad=1, bias=b).view(64, 128, 1024)
    part = K.conv_gemm(x[16:48].reshape(-1, 256), w, 1024, T=128, taps=3, pad=1, bias=b).view(32, 128, 1024)
    assert torch.equal(full[16:48], part)
