"""Host logic of optispeech_amd/precision.py (no GPU): which scope lets an f32 GEMM take the split-bf16 kernels, and that every scope
restores the state it found -- the mode is consulted again by every backward pass, so a leaked scope would silently change kernels."""
import pytest

from optispeech_amd import precision as P


@pytest.fixture(autouse=True)
def _restore():
    keep = (P.get_precision(), P._split["v"], P._fwd_parity["v"], P._split_wgrad["v"], P._split_bwd["v"])
    try:
        yield
    finally:
        P.set_precision(keep[0])
        P._split["v"], P._fwd_parity["v"], P._split_wgrad["v"], P._split_bwd["v"] = keep[1:]


def test_split_only_in_mixed_mode_outside_the_index_path():
    P.set_f32_split(True)
    for mode, want in (("f32", False), ("bf16", False), ("mixed", True)):
        P.set_precision(mode)
        assert P.f32_split() is want and P.f32_split("wgrad") is want
        with P.index_path():
            assert not P.f32_split()                       # index-critical: the exact kernels in every mode
            P.leave_index_path()
            assert P.f32_split() is want                   # the rest of the block is continuous again
        assert P.f32_split() is want
    P.set_precision("mixed")
    with P.disc_scope():                                   # the discriminator stacks see bf16: no f32 GEMM, no split
        assert P.is_bf16() and not P.f32_split()
        with P.generator_scope():                          # the spectral losses inside it are the generator's again
            assert P.f32_split()
    P.set_f32_split(False)
    assert not P.f32_split()


def test_per_site_switches():
    P.set_precision("mixed")
    P.set_f32_split(True)
    P._split_wgrad["v"] = False
    assert P.f32_split() and not P.f32_split("wgrad")
    P._split_wgrad["v"] = True
    P._split_bwd["v"] = False
    assert P.f32_split() and P.f32_split("wgrad")          # not inside an autograd backward pass here
    import torch
    seen = []

    class _F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            seen.append(P.f32_split())
            return g * 2
    x = torch.ones(2, requires_grad=True)
    _F.apply(x).sum().backward()
    assert seen == [False]                                 # OSP_F32_SPLIT_BWD=0: exact kernels inside a backward pass
    P._split_bwd["v"] = True
    _F.apply(x).sum().backward()
    assert seen == [False, True]


def test_parity_forward_scope_is_bf16_only_opt_in_and_restores():
    P.set_precision("bf16")
    P.set_forward_parity(False)
    with P.parity_forward():
        assert P.is_bf16() and not P.f32_split()           # off by default: the scope is a no-op
    P.set_forward_parity(True)
    with P.parity_forward():
        assert not P.is_bf16() and P.f32_split()           # the generator's forward sees the parity mode's kernels ...
        assert P.get_precision() == "f32"                  # ... and the EFFECTIVE mode agrees with is_bf16() ...
        assert P.configured_precision() == "bf16" and P.signature()[0] == "bf16"     # ... while tape / graph keys name the configured mode
        with P.index_path():
            assert not P.f32_split()
        with P.parity_forward():                           # nesting is a no-op
            assert P.f32_split()
        assert P.f32_split()
    assert P.is_bf16() and not P.f32_split() and P.get_precision() == "bf16" == P.configured_precision()
    with pytest.raises(RuntimeError):
        with P.parity_forward():
            raise RuntimeError("forward failed")
    assert P.is_bf16() and not P._in_fwd["v"]              # restored on an exception too
    for mode in ("f32", "mixed"):                          # only the bf16 mode has anything to upgrade
        P.set_precision(mode)
        with P.parity_forward():
            assert not P._in_fwd["v"] and P.get_precision() == mode


def test_signature_names_every_kernel_choosing_switch():
    P.set_precision("mixed")
    base = P.signature()
    assert base[0] == "mixed"
    for setter in (lambda: P.set_f32_split(not P._split["v"]), lambda: P._split_wgrad.__setitem__("v", not P._split_wgrad["v"]),
                   lambda: P._split_bwd.__setitem__("v", not P._split_bwd["v"]), lambda: P.set_forward_parity(not P._fwd_parity["v"])):
        setter()
        assert P.signature() != base                      # a tape / graph key built from it cannot be replayed under other switches
        setter()
        assert P.signature() == base
    P.set_precision("bf16")
    P.set_forward_parity(True)
    with P.parity_forward():
        assert P.signature()[0] == "bf16"                 # the configured mode, also inside the scope
