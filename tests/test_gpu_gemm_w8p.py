"""The phased 8-wave conv-GEMM kernels (csrc/gemm_bf16_w8p.hip, gemm_bf16_w8q.hip: the DiscriminatorP 512 -> 1024 / 1024 -> 1024 layers,
reference vocoder/wavenext/disc/_discriminators.py:51-60) against the lock-step kernel they replace and against torch.

All three main loops accumulate the same k-slabs in the same order on the same MFMA, so their outputs must be EQUAL BIT FOR BIT:
any difference is a synchronisation defect (a fragment read racing an LDS-DMA write), not rounding.  Every case is therefore also
launched repeatedly (a race shows as a launch that differs from the others)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16_mode_and_env():
    from optispeech_amd import precision
    old = {k: os.environ.get(k) for k in ("OSP_GEMM_W8", "OSP_GEMM_W8P")}
    prev = precision.get_precision()
    precision.set_precision("bf16")
    yield
    precision.set_precision(prev)
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _variants(fn, reps=6):
    """fn() under the lock-step kernel (0) and the phased kernels (1, 2, 3 = tap reuse where it applies, else 2); returns the lock-step output after asserting that every
    launch of every variant equals it bitwise."""
    os.environ["OSP_GEMM_W8"] = "2"                 # take the 8-wave family whatever the problem size (short K, few tiles)
    os.environ["OSP_GEMM_W8P"] = "0"
    base = fn().clone()
    for v in ("1", "2", "3"):
        os.environ["OSP_GEMM_W8P"] = v
        for r in range(reps):
            out = fn()
            assert torch.equal(out, base), f"variant {v}, launch {r}: {(out.float() - base.float()).abs().max().item():.3e} max difference"
    return base


@pytest.mark.parametrize("U,T,cin,n,taps,stride", [
    (128, 102, 1024, 1024, 5, 1),        # the layer-5 forward of the step (204 tiles)
    (64, 304, 512, 1024, 5, 3),          # layer 4, stride 3
    (37, 53, 256, 520, 3, 1),            # ragged: M = 1961 (not a multiple of 256), N = 520 (partial column tile), short rows
    (9, 700, 128, 256, 7, 2),            # long rows, 7 taps, stride 2
    (300, 19, 1024, 768, 5, 1),          # period-11 shape: rows shorter than the tap window's reach, three column tiles
])
def test_conv_forward_bit_identical_and_vs_torch(U, T, cin, n, taps, stride):
    from optispeech_amd import kernels as K
    torch.manual_seed(U + T)
    pad = taps // 2
    Tout = (T + 2 * pad - taps) // stride + 1
    M = U * Tout
    a = torch.randn(U * T, cin, device="cuda").bfloat16()
    w = (torch.randn(n, taps, cin, device="cuda") * (1.0 / (taps * cin) ** 0.5)).bfloat16()
    bias = torch.randn(n, device="cuda")
    fn = lambda: K.conv_gemm_bf16(a, w, n, M=M, Trows=Tout, Tin=T, cin=cin, taps=taps, a_step=stride, a_off=-pad, out_bf16=True,  # noqa: E731
                                  epi=K.EPI_LRELU, bias=bias, slope=0.1)
    out = _variants(fn)
    ref = F.conv1d(a.float().view(U, T, cin).transpose(1, 2), w.float().permute(0, 2, 1).contiguous(), bias, stride=stride, padding=pad)
    ref = F.leaky_relu(ref, 0.1).transpose(1, 2).reshape(M, n)
    assert (out.float() - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()


@pytest.mark.parametrize("U,T,cin,n", [
    (150, 17, 64, 256),          # the shortest utterances the panel takes (16 of them in a tile), ONE channel block
    (77, 23, 128, 512),          # two channel blocks (one trip of the ten-tile loop, no tail), ragged M = 1771
    (40, 131, 192, 256),         # three channel blocks (loop + tail), utterances longer than half a tile
    (11, 1000, 320, 768),        # utterances longer than a tile: tiles that start in the middle of one
])
def test_tap_reuse_kernel_cases(U, T, cin, n):
    """gemm_bf16_w8r.hip (variant 3): 5 taps, stride 1 -- forward (tap step +1) and the stride-1 dgrad (tap step -1), shapes that move the
    utterance boundaries through the panel; bit-identical to the per-tap kernels."""
    from optispeech_amd import kernels as K, disc_ops as D
    torch.manual_seed(U + T + cin)
    M = U * T
    a = torch.randn(M, cin, device="cuda").bfloat16()
    w = (torch.randn(n, 5, cin, device="cuda") * (1.0 / (5 * cin) ** 0.5)).bfloat16()
    out = _variants(lambda: K.conv_gemm_bf16(a, w, n, M=M, Trows=T, Tin=T, cin=cin, taps=5, a_step=1, a_off=-2, out_bf16=False))
    ref = F.conv1d(a.float().view(U, T, cin).transpose(1, 2), w.float().permute(0, 2, 1).contiguous(), padding=2).transpose(1, 2).reshape(M, n)
    assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    # the stride-1 dgrad of the same layer: dx (U, 1, T, cin) from dy (U, 1, T, n)
    dy = torch.randn(U, 1, T, n, device="cuda").bfloat16()
    w4 = w.view(n, 1, 5, cin)
    wt = w4.permute(3, 1, 2, 0).contiguous()
    dx = _variants(lambda: D.conv2d_dgrad(dy, wt, 1, T, 1, 5, 1, 1, 0, 2, out_bf16=False))
    xf = torch.zeros(U, cin, 1, T, device="cuda", requires_grad=True)
    g, = torch.autograd.grad(F.conv2d(xf, w4.float().permute(0, 3, 1, 2), padding=(0, 2)), xf, dy.float().permute(0, 3, 1, 2))
    ref = g.permute(0, 2, 3, 1)
    assert (dx - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("seed", range(12))
def test_random_conv_shapes(seed):
    """Seeded random problems (utterance length, count, channels, taps 1-9, stride 1-3, ragged N): every 8-wave main loop gives the
    lock-step kernel's bits, and those agree with torch."""
    import random
    from optispeech_amd import kernels as K
    rnd = random.Random(1000 + seed)
    taps = rnd.choice([1, 3, 5, 5, 7, 9]); stride = rnd.choice([1, 1, 2, 3]); pad = taps // 2
    cin = 64 * rnd.randint(1, 6); n = 8 * rnd.randint(33, 130)
    T = rnd.randint(max(17, taps), 400); Tout = (T + 2 * pad - taps) // stride + 1
    U = max(1, rnd.randint(600, 9000) // Tout)
    M = U * Tout
    torch.manual_seed(seed)
    a = torch.randn(U * T, cin, device="cuda").bfloat16()
    w = (torch.randn(n, taps, cin, device="cuda") * (1.0 / (taps * cin) ** 0.5)).bfloat16()
    bias = torch.randn(n, device="cuda")
    out = _variants(lambda: K.conv_gemm_bf16(a, w, n, M=M, Trows=Tout, Tin=T, cin=cin, taps=taps, a_step=stride, a_off=-pad, out_bf16=False, bias=bias), reps=3)
    ref = F.conv1d(a.float().view(U, T, cin).transpose(1, 2), w.float().permute(0, 2, 1).contiguous(), bias, stride=stride, padding=pad)
    ref = ref.transpose(1, 2).reshape(M, n)
    assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item(), (U, T, cin, n, taps, stride)


@pytest.mark.parametrize("K_", [64, 128, 192, 256, 320, 448])
def test_short_reductions_prologue_and_tail(K_):
    """nk = 1 .. 7 k-slabs: the prologue requests more units than exist, the tail waits count down (vmcnt(8) .. vmcnt(0))."""
    from optispeech_amd import kernels as K
    torch.manual_seed(K_)
    M, N = 128 * 256, 512
    a = torch.randn(M, K_, device="cuda").bfloat16()
    w = (torch.randn(N, K_, device="cuda") * 0.1).bfloat16()
    out = _variants(lambda: K.conv_gemm_bf16(a, w, N, M=M, Trows=M, Tin=M, cin=K_, out_bf16=False))
    ref = a.float() @ w.float().t()
    assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("cin,cout,W,s", [(512, 1024, 304, 3), (1024, 1024, 102, 1), (256, 512, 77, 2)])
def test_fused_phase_dgrad_bit_identical_and_vs_torch(cin, cout, W, s):
    """The strided-conv dgrad with all output phases in one launch (per-phase tap subsets and weight offsets), LeakyReLU' epilogue
    with the feature-matching addend."""
    from optispeech_amd import disc_ops as D
    torch.manual_seed(cin + W)
    U = 96
    Wo = (W + 4 - 5) // s + 1
    x = torch.randn(U, 1, W, cin, device="cuda").bfloat16()
    extra = (torch.randn(U, 1, W, cin, device="cuda") * 0.1).bfloat16()
    w = (torch.randn(cout, 1, 5, cin, device="cuda") * 0.02).bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    dy = torch.randn(U, 1, Wo, cout, device="cuda").bfloat16()
    fn = lambda: D.conv2d_dgrad(dy, wt, 1, W, 1, 5, 1, s, 0, 2, lrelu_y=x, extra=extra, slope=0.1, out_bf16=False)   # noqa: E731
    out = _variants(fn)
    xf = x.float().permute(0, 3, 1, 2).requires_grad_(True)            # (U, C, 1, W)
    y = F.conv2d(xf, w.float().permute(0, 3, 1, 2), stride=(1, s), padding=(0, 2))
    g, = torch.autograd.grad(y, xf, dy.float().permute(0, 3, 1, 2))
    g = (g.permute(0, 2, 3, 1) + extra.float())
    ref = torch.where(x.float() > 0, g, 0.1 * g)
    assert (out - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()


def test_batched_launch_and_f32_rows():
    """grid z = batch (per-batch operand strides move the descriptor base), f32 destination, accumulate."""
    from optispeech_amd import kernels as K
    torch.manual_seed(5)
    Bt, M, N, Kd = 3, 2048, 512, 2304
    a = torch.randn(Bt, M, Kd, device="cuda").bfloat16()
    w = (torch.randn(Bt, N, Kd, device="cuda") * 0.05).bfloat16()
    c0 = torch.randn(Bt, M, N, device="cuda")

    def fn():
        out = c0.clone()
        K.conv_gemm_bf16(a, w, N, M=M, Trows=M, Tin=M, cin=Kd, out=out, ldc=N, accumulate=True, batch=Bt,
                         batch_strides=(M * Kd, N * Kd, M * N, 0))
        return out
    out = _variants(fn)
    ref = c0 + torch.einsum("bmk,bnk->bmn", a.float(), w.float())
    assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_default_is_the_rotating_schedule_kernel():
    """The dispatcher's default for the step's DiscriminatorP shapes is gemm_bf16_w8q.hip (the library names the symbol it launched:
    osp_kernel_note_host, what bench.py's roofline pass reads)."""
    import ctypes
    from optispeech_amd import kernels as K, _lib
    os.environ.pop("OSP_GEMM_W8", None); os.environ.pop("OSP_GEMM_W8P", None)
    note = _lib.lib().cdll.osp_kernel_note_host
    note.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    buf, fl = ctypes.create_string_buffer(128), ctypes.c_double(0.0)
    U, T, cin, n = 128, 102, 1024, 1024
    a = torch.randn(U * T, cin, device="cuda").bfloat16(); w = torch.randn(n, 5, cin, device="cuda").bfloat16()
    note(buf, 128, ctypes.byref(fl))
    K.conv_gemm_bf16(a, w, n, M=U * T, Trows=T, Tin=T, cin=cin, taps=5, a_off=-2, out_bf16=True)
    note(buf, 128, ctypes.byref(fl))
    torch.cuda.synchronize()
    assert buf.value.decode() == "conv_gemm_bf16_glds8q_kernel"
    assert fl.value == 2.0 * U * T * 5 * cin * n
