"""The phased 8-wave weight-gradient kernel (csrc/wgrad_bf16_tr8q.hip: DiscriminatorP 512 -> 1024 / 1024 -> 1024 weight gradients,
reference vocoder/wavenext/disc/_discriminators.py:51-60 under autograd) against the lock-step kernel it replaces and against torch.

Both kernels add the same frames in the same order into the same accumulators: with ONE frame split (no atomics between workgroups)
the results must be equal bit for bit, launch after launch -- a difference is a synchronisation defect.  With several frame splits the
partial sums meet in f32 atomics (order-dependent in the last bit, in both kernels): compared with a tolerance there."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _env():
    from optispeech_amd import precision
    old = os.environ.get("OSP_WGRAD_W8Q")
    prev = precision.get_precision()
    precision.set_precision("bf16")
    yield
    precision.set_precision(prev)
    if old is None:
        os.environ.pop("OSP_WGRAD_W8Q", None)
    else:
        os.environ["OSP_WGRAD_W8Q"] = old


def _wgrad(dy, x, U, Tout, Tin, cout, cin, st):
    from optispeech_amd import kernels as K
    dw = torch.zeros(cout, 5, cin, device="cuda"); db = torch.zeros(cout, device="cuda")
    K.conv_wgrad_bf16(dy.view(U * Tout, cout), x.view(U * Tin, cin), dw, db, M=U * Tout, Trows=Tout, Tin=Tin, n=cout, cin=cin, taps=5, pad=2, x_step=st)
    return dw, db


def _symbol(fn):
    import ctypes
    from optispeech_amd import _lib
    note = _lib.lib().cdll.osp_kernel_note_host
    note.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    buf, fl = ctypes.create_string_buffer(128), ctypes.c_double(0.0)
    note(buf, 128, ctypes.byref(fl))
    fn()
    note(buf, 128, ctypes.byref(fl))
    return buf.value.decode()


@pytest.mark.parametrize("U,Tin,cin,cout,st,bitwise", [
    (90, 97, 2048, 1024, 1, True),       # 160 tiles: one frame split, no atomics between workgroups; M = 8730 (ragged last slab)
    (600, 19, 2048, 1024, 1, True),      # rows shorter than a 16-frame unit: several utterance wraps per request
    (128, 102, 1024, 1024, 1, False),    # the step's layer 5 (80 tiles -> 3 frame splits)
    (128, 304, 512, 1024, 3, False),     # layer 4, stride 3 (40 tiles -> 6 splits)
    (704, 19, 1024, 1024, 1, False),     # period 11
])
def test_phased_equals_lock_step_and_torch(U, Tin, cin, cout, st, bitwise):
    torch.manual_seed(U + Tin)
    Tout = (Tin + 4 - 5) // st + 1
    x = torch.randn(U, Tin, cin, device="cuda").bfloat16()
    dy = (torch.randn(U, Tout, cout, device="cuda") * 0.1).bfloat16()
    os.environ["OSP_WGRAD_W8Q"] = "0"
    assert _symbol(lambda: _wgrad(dy, x, U, Tout, Tin, cout, cin, st)) == "conv_wgrad_bf16_tr8_kernel"
    dw0, db0 = _wgrad(dy, x, U, Tout, Tin, cout, cin, st)
    os.environ["OSP_WGRAD_W8Q"] = "1"
    assert _symbol(lambda: _wgrad(dy, x, U, Tout, Tin, cout, cin, st)) == "conv_wgrad_bf16_tr8q_kernel"
    scale = dw0.abs().max().item()
    for r in range(5):
        dw1, db1 = _wgrad(dy, x, U, Tout, Tin, cout, cin, st)
        if bitwise:
            assert torch.equal(dw1, dw0), f"launch {r}: max difference {(dw1 - dw0).abs().max().item():.3e}"
        else:
            assert (dw1 - dw0).abs().max().item() <= 2e-6 * scale
        assert (db1 - db0).abs().max().item() <= 2e-6 * db0.abs().max().item()
    # against torch (f32 accumulation of the same bf16 operands)
    xf = x.float().transpose(1, 2)
    w = torch.zeros(cout, cin, 5, device="cuda", requires_grad=True)
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    y = F.conv1d(xf, w, b, stride=st, padding=2)
    gw, gb = torch.autograd.grad(y, (w, b), dy.float().transpose(1, 2))
    ref = gw.permute(0, 2, 1)
    assert (dw1 - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert (db1 - gb).abs().max().item() <= 2e-3 * gb.abs().max().item()
