"""Small-problem conv-GEMM kernel (csrc/gemm_bf16_small.hip: 64x64 tiles, LDS-DMA ring, bf16 or f32 A operand) against torch on the
bf16-rounded operands: the generator-side shapes of the training step and the edges (one / two / three slabs, ragged M and N,
conv taps with zero padding, fused epilogues)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("a32", [False, True])
@pytest.mark.parametrize("M,N,Kd", [(2048, 384, 1152), (2048, 1152, 384), (4096, 256, 1024), (1000, 200, 64), (130, 72, 128),
                                    (64, 65, 192), (4096, 1024, 256)])
def test_pointwise_matches_torch(a32, M, N, Kd):
    from optispeech_amd import kernels as K
    a = rnd(M, Kd, seed=1)
    w = rnd(N, Kd, seed=2, scale=Kd ** -0.5)
    bias = rnd(N, seed=3)
    a_in = a if a32 else a.to(torch.bfloat16)
    want = bf(a) @ bf(w).t() + bias
    got = K.conv_gemm_bf16(a_in, w.to(torch.bfloat16), N, M=M, Trows=M, Tin=M, cin=Kd, bias=bias)
    assert got.dtype == torch.float32
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    # bf16 output + ReLU
    got16 = K.conv_gemm_bf16(a_in, w.to(torch.bfloat16), N, M=M, Trows=M, Tin=M, cin=Kd, bias=bias, epi=K.EPI_RELU, out_bf16=True)
    torch.testing.assert_close(got16.float(), torch.relu(want), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("a32", [False, True])
@pytest.mark.parametrize("B,T,Cin,Cout,taps", [(32, 128, 256, 256, 5), (32, 128, 384, 384, 3), (3, 37, 64, 128, 7), (32, 64, 256, 384, 7)])
def test_conv_taps_match_conv1d(a32, B, T, Cin, Cout, taps):
    from optispeech_amd import kernels as K
    pad = (taps - 1) // 2
    x = rnd(B, T, Cin, seed=4)
    w = rnd(Cout, taps, Cin, seed=5, scale=(taps * Cin) ** -0.5)           # native tap-major layout
    bias = rnd(Cout, seed=6)
    want = F.conv1d(bf(x).transpose(1, 2), bf(w).permute(0, 2, 1).contiguous(), bias, padding=pad).transpose(1, 2).reshape(B * T, Cout)
    a_in = (x if a32 else x.to(torch.bfloat16)).view(B * T, Cin)
    got = K.conv_gemm_bf16(a_in, w.to(torch.bfloat16), Cout, M=B * T, Trows=T, Tin=T, cin=Cin, taps=taps, a_off=-pad, bias=bias)
    torch.testing.assert_close(got, want, rtol=3e-4, atol=3e-4)


def test_gelu_epilogue_with_saved_preactivation_and_its_backward():
    from optispeech_amd import kernels as K
    M, C, I = 2048, 384, 1152
    h = rnd(M, C, seed=7).to(torch.bfloat16)
    W1 = rnd(I, C, seed=8, scale=C ** -0.5).to(torch.bfloat16)
    b1 = rnd(I, seed=9)
    u = torch.empty((M, I), device=DEV, dtype=torch.bfloat16)
    g = K.conv_gemm_bf16(h, W1, I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=b1, aux_out=u, out_bf16=True)
    pre = h.float() @ W1.float().t() + b1
    torch.testing.assert_close(u.float(), pre, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(g.float(), F.gelu(pre), rtol=1e-2, atol=1.5e-2)
    # du = (dy @ W) * gelu'(u)     (A operand f32: the a32 kernel; u read back as aux_in)
    dy = rnd(M, C, seed=10)
    Wt = rnd(I, C, seed=11, scale=C ** -0.5).to(torch.bfloat16)
    du = K.conv_gemm_bf16(dy, Wt, I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU_BWD, aux_in=u, out_bf16=True)
    uu = u.float().requires_grad_(True)
    F.gelu(uu).backward(bf(dy) @ Wt.float().t())
    torch.testing.assert_close(du.float(), uu.grad, rtol=2e-2, atol=2e-2)


def test_cached_parameter_packs_follow_the_parameter():
    """conv_gemm(w_param=...) keeps bf16 packs on the Parameter: an in-place update or a new optimizer epoch must refresh them."""
    from optispeech_amd import kernels as K, precision, values
    keep = precision.get_precision() if hasattr(precision, "get_precision") else None
    precision.set_precision("bf16")
    try:
        M, Cin, Cout, taps = 4096, 256, 256, 5
        x = rnd(M, Cin, seed=12)
        w = torch.nn.Parameter(rnd(Cout, taps, Cin, seed=13, scale=0.03))
        y0 = K.conv_gemm(x, w, Cout, T=128, taps=taps, pad=2, w_param=w)
        y1 = K.conv_gemm(x, w, Cout, T=128, taps=taps, pad=2, w_param=w)
        assert torch.equal(y0, y1)
        with torch.no_grad():
            w.mul_(2.0)                                   # bumps the version counter
        y2 = K.conv_gemm(x, w, Cout, T=128, taps=taps, pad=2, w_param=w)
        torch.testing.assert_close(y2, 2.0 * y0, rtol=1e-5, atol=1e-5)
        w.data = w.data * 0.5                             # new storage, same Parameter
        y3 = K.conv_gemm(x, w, Cout, T=128, taps=taps, pad=2, w_param=w)
        torch.testing.assert_close(y3, y0, rtol=1e-5, atol=1e-5)
    finally:
        if keep is not None:
            precision.set_precision(keep)
