"""csrc/conv2d_panel.hip -- the tap-reuse ("kw-panel") kernel of the 64 <- 64 DiscriminatorR layers and their fused-phase input
gradients -- against torch's f32 conv2d on the same bf16-rounded operands.

Reference op: Conv2d(64, 64, (5, 3) | (3, 3), stride (2, 1) | (2, 2)) of DiscriminatorR
(optispeech/model/vocoder/wavenext/disc/_discriminators.py:153-158) and its autograd input gradient.  Both sides multiply the SAME
bf16 numbers and accumulate in f32, so the bound is accumulation order only: 2e-5 of the tensor's scale for f32 outputs (one bf16
rounding, 4e-3, for bf16 outputs).  Shapes: the real bin / frame counts of the three resolutions' layers 1-4 (odd line lengths 129, 65,
33, 17, 9: many line segments inside one 128-row tile, the last tile partial, utterance boundaries inside a tile), every stride
combination, both tap directions (forward: +1, dgrad: -1), 1-tap dgrad phases.  Every case asserts that the panel kernel is the one
that ran (the library's dispatcher note), not the glds kernel it replaces."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=["1", "2"], ids=["per-tap-barrier", "pipelined"])
def _panel_on(request):
    """The panel kernel is opt-in (OSP_N64_PANEL=1: a barrier per tap; 2: the software-pipelined loop on a four-stage weight ring; the
    launcher reads the variable per call)."""
    import os
    keep = os.environ.get("OSP_N64_PANEL")
    os.environ["OSP_N64_PANEL"] = request.param
    yield
    if keep is None:
        os.environ.pop("OSP_N64_PANEL", None)
    else:
        os.environ["OSP_N64_PANEL"] = keep


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def bfr(x):
    return x.to(torch.bfloat16).float()


def _note():
    from optispeech_amd import _lib
    f = _lib.lib().cdll.osp_kernel_note_host
    f.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    buf, fl = ctypes.create_string_buffer(128), ctypes.c_double(0.0)
    f(buf, 128, ctypes.byref(fl))
    return buf.value.decode()


def _fits(Wo, KW, sw):
    """csrc/conv2d_panel.hip panel_fits: the 288-row panel holds a 128-row tile's segments (very short lines -- 3 bins -- do not)."""
    return Wo > 0 and sw <= KW <= 8 and 128 * sw + ((128 + Wo - 2) // Wo + 1) * (KW - sw) + KW <= 288


def _err(got, want):
    got, want = got.detach().float().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


# (U, H, W, (KH, KW, sh, sw, ph, pw)) in the package's (frames, bins) orientation (disc_ops.MRD_SPEC)
L1, L2, L3, L4 = (3, 5, 1, 2, 1, 2), (3, 5, 2, 2, 1, 2), (3, 3, 1, 2, 1, 1), (3, 3, 2, 2, 1, 1)
CASES = [(3, 17, 257, L1), (2, 17, 129, L2), (3, 9, 65, L3), (5, 9, 33, L4),          # n_fft 1024
         (2, 9, 513, L1), (3, 5, 129, L3), (7, 5, 65, L4),                            # n_fft 2048
         (2, 33, 129, L1), (2, 33, 65, L2), (4, 17, 33, L3), (9, 17, 17, L4),         # n_fft 512
         (1, 3, 9, L4), (1, 1, 5, L3), (2, 4, 300, (3, 5, 1, 1, 1, 2)), (2, 6, 40, (1, 3, 1, 2, 0, 1))]


@pytest.mark.parametrize("U,H,W,spec", CASES)
@pytest.mark.parametrize("out_bf16", [False, True])
def test_panel_forward(U, H, W, spec, out_bf16):
    from optispeech_amd import disc_ops as D, kernels as K
    KH, KW, sh, sw, ph, pw = spec
    x = bfr(rnd(U, 64, H, W, seed=11))
    w = bfr(rnd(64, 64, KH, KW, seed=12, scale=1.0 / np.sqrt(64 * KH * KW)))
    b = rnd(64, seed=13)
    want = F.leaky_relu(F.conv2d(x, w, b, stride=(sh, sw), padding=(ph, pw)), 0.1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)
    wn = K.cast_bf16(w.permute(0, 2, 3, 1).contiguous().to(DEV))
    _note()
    got = D.conv2d_fwd(xg, wn, b.to(DEV), *spec, 0.1, out_bf16)
    sym = _note()
    torch.cuda.synchronize()
    if _fits(((W + 2 * pw - KW) // sw + 1), KW, sw):
        assert sym == "conv2d_panel_n64_kernel", sym
    assert _err(got.permute(0, 3, 1, 2), want) < (4e-3 if out_bf16 else 2e-5)


@pytest.mark.parametrize("U,H,W,spec", CASES)
def test_panel_fused_phase_dgrad(U, H, W, spec):
    """The input gradient with the previous layer's LeakyReLU' and the feature-matching addend fused (EPI_LRELU_BWD): all output
    phases of the strided conv in one launch, each a dense conv over dy with a sub-sampled kernel walked BACKWARDS (tap step -1)."""
    from optispeech_amd import disc_ops as D, kernels as K
    KH, KW, sh, sw, ph, pw = spec
    x = bfr(rnd(U, 64, H, W, seed=21)).requires_grad_(True)
    w = bfr(rnd(64, 64, KH, KW, seed=22, scale=1.0 / np.sqrt(64 * KH * KW)))
    y = F.conv2d(x, w, None, stride=(sh, sw), padding=(ph, pw))
    dy = bfr(rnd(*y.shape, seed=23))
    y.backward(dy)
    yprev = bfr(rnd(U, 64, H, W, seed=24))                                   # forward output of the layer below (sign decides the slope)
    extra = bfr(rnd(U, 64, H, W, seed=25, scale=0.1))                        # its feature-matching gradient
    want = (x.grad + extra) * torch.where(yprev > 0, 1.0, 0.1)
    cl = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)      # noqa: E731
    wn = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    _note()
    dx = D.conv2d_dgrad(cl(dy), D.transpose_weight2d(wn), H, W, *spec, lrelu_y=cl(yprev), extra=cl(extra))
    sym = _note()
    torch.cuda.synchronize()
    if all(_fits((W - rw + sw - 1) // sw, (KW - (rw + pw) % sw + sw - 1) // sw, 1) for rw in range(sw)):
        assert sym == "conv2d_panel_n64_kernel", sym
    assert _err(dx.permute(0, 3, 1, 2), want) < 2e-5
    plain = D.conv2d_dgrad(cl(dy), D.transpose_weight2d(wn), H, W, *spec, out_bf16=True)
    assert _err(plain.permute(0, 3, 1, 2), x.grad) < 4e-3


def test_panel_is_bit_reproducible_and_matches_the_per_tap_kernel_elementwise():
    """Two runs are bit-identical (no atomics, fixed order), and so is the result of the per-tap glds kernel the panel kernel
    replaces: same products, same k order (kernel row outer, tap, channel), same f32 accumulation."""
    import os
    import subprocess
    import sys
    from optispeech_amd import disc_ops as D, kernels as K
    x = bfr(rnd(4, 17, 129, 64, seed=31)).to(DEV).to(torch.bfloat16)
    wn = K.cast_bf16(rnd(64, 3, 5, 64, seed=32, scale=0.05).to(DEV))
    b = rnd(64, seed=33).to(DEV)
    a = D.conv2d_fwd(x, wn, b, *L1, 0.1, False)
    c = D.conv2d_fwd(x, wn, b, *L1, 0.1, False)
    assert torch.equal(a, c)
    code = ("import torch, sys; sys.path.insert(0, %r); from tests.test_gpu_conv2d_panel import *; from tests.test_gpu_conv2d_panel import _note; from optispeech_amd import disc_ops as D, kernels as K\n"
            "x = bfr(rnd(4, 17, 129, 64, seed=31)).to(DEV).to(torch.bfloat16); wn = K.cast_bf16(rnd(64, 3, 5, 64, seed=32, scale=0.05).to(DEV)); b = rnd(64, seed=33).to(DEV)\n"
            "_note(); y = D.conv2d_fwd(x, wn, b, *L1, 0.1, False); assert _note() == 'conv_gemm_bf16_glds_n64_kernel'\n"
            "torch.save(y.cpu(), sys.argv[1])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = "/tmp/osp_panel_ref.pt"
    env = dict(os.environ, OSP_N64_PANEL="0")
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
    ref = torch.load(out)
    assert _err(a, ref) < 1e-6                                  # (the k order differs by nothing, the MFMA grouping is the same)
