"""One-kernel ConvNeXt block MLP of the no-grad path (csrc/mlp_fused.hip, osp_convnext_mlp_fused) against

  * a float64 restatement of generator/modules/convnext.py:39-46 on the operands the kernel sees (h, W1, W2 rounded to bf16, the
    GELU output rounded to bf16 as in the performance mode's two-launch path), and
  * the two-launch path itself (conv_gemm_bf16 GELU epilogue -> conv_gemm_bf16 scale/residual/mask epilogue), which is what the
    golden synthesise fixtures were accepted on.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _case(M, C, I, seed, mask):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(M, C, generator=g)
    x = torch.randn(M, C, generator=g)
    W1 = torch.randn(I, C, generator=g) / C ** 0.5
    W2 = torch.randn(C, I, generator=g) / I ** 0.5
    b1 = torch.randn(I, generator=g) * 0.2
    b2 = torch.randn(C, generator=g) * 0.2
    gamma = torch.randn(C, generator=g) * 0.5
    rowmask = (torch.rand(M, generator=g) > 0.25).float() if mask else None
    rowscale = ((torch.rand(M, generator=g) > 0.2).float() / 0.8) if mask else None      # DropPath factors (0 or 1 / keep)
    return h, x, W1, W2, b1, b2, gamma, rowmask, rowscale


def _reference(h, x, W1, W2, b1, b2, gamma, rowmask, rowscale):
    u = _bf16(h) @ _bf16(W1).T + b1.double()
    gl = _bf16(0.5 * u * (1.0 + torch.erf(u / 2.0 ** 0.5)))
    y = gamma.double() * (gl @ _bf16(W2).T + b2.double())
    y = x.double() + (y * rowscale.double()[:, None] if rowscale is not None else y)
    return y * rowmask.double()[:, None] if rowmask is not None else y


@pytest.mark.parametrize("C,I", [(384, 1152), (256, 1024), (384, 128), (256, 256)])
@pytest.mark.parametrize("M,mask", [(1, False), (31, True), (128, False), (129, True), (1000, True), (4133, False)])
def test_fused_mlp_matches_restatement_and_two_launch_path(C, I, M, mask):
    from optispeech_amd import kernels as K, precision
    h, x, W1, W2, b1, b2, gamma, rowmask, rowscale = _case(M, C, I, 1000 * C + I + M, mask)
    ref = _reference(h, x, W1, W2, b1, b2, gamma, rowmask, rowscale)
    d = lambda t: None if t is None else t.to(DEV)
    hb = h.to(DEV).to(torch.bfloat16)
    W1p, W2p = torch.nn.Parameter(W1.to(DEV)), torch.nn.Parameter(W2.to(DEV))
    precision.set_precision("bf16")
    try:
        y = K.convnext_mlp_fused(hb, W1p, d(b1), W2p, d(b2), d(gamma), d(x), d(rowmask), d(rowscale))
        gg = K.conv_gemm_bf16(hb, K.param_bf16(W1p), I, M=M, Trows=M, Tin=M, cin=C, epi=K.EPI_GELU, bias=d(b1), out_bf16=True)
        y2 = K.conv_gemm_bf16(gg, K.param_bf16(W2p), C, M=M, Trows=M, Tin=M, cin=I, epi=K.EPI_SCALE_RES_MASK, bias=d(b2),
                              gamma=d(gamma), res=d(x), rowmask=d(rowmask), rowscale=d(rowscale))
        torch.cuda.synchronize()
    finally:
        precision.set_precision("f32")
    y, y2 = y.cpu().double(), y2.cpu().double()
    assert torch.isfinite(y).all()
    scale = ref.abs().max().item()
    # the restatement rounds gelu(u) to bf16 from float64 u; the kernels from f32 accumulations of bf16 products: a different
    # rounding of a few hidden units per row moves an output by <= |gamma W2| * 2^-9 * |g|
    assert (y - ref).abs().max().item() <= 4e-3 * scale, ((y - ref).abs().max().item(), scale)
    assert (y - y2).abs().max().item() <= 4e-3 * scale, ((y - y2).abs().max().item(), scale)
    # aggregate agreement is much tighter than the worst element
    assert ((y - ref) ** 2).mean().sqrt().item() <= 3e-4 * scale
    if rowmask is not None:
        assert (y[rowmask == 0] == 0).all()


@pytest.mark.parametrize("C,I", [(384, 1152), (256, 1024)])
@pytest.mark.parametrize("M", [64, 8200, 20000, 40000, 49152 + 77])
def test_split_mode_tail_equals_full_mode(C, I, M, monkeypatch):
    """Round 6: the last, at most half-full round of 128-row blocks goes out as 64-row split-mode workgroups (two waves per row tile,
    each half of a chunk's hidden units, partial output tiles summed through LDS).  Same S^T, same GELU roundings; only the f32
    summation order of the output differs from the full mode (OSP_MLP_SPLIT=0).  M covers: everything split (64, 8200), nothing
    split (20000: 157 blocks), a full round + split tail (40000, 49229)."""
    from optispeech_amd import kernels as K, precision
    h, x, W1, W2, b1, b2, gamma, rowmask, rowscale = _case(M, C, I, 7 * C + I + M, True)
    d = lambda t: None if t is None else t.to(DEV)
    hb = h.to(DEV).to(torch.bfloat16)
    W1p, W2p = torch.nn.Parameter(W1.to(DEV)), torch.nn.Parameter(W2.to(DEV))
    args = (hb, W1p, d(b1), W2p, d(b2), d(gamma), d(x), d(rowmask), d(rowscale))
    precision.set_precision("bf16")
    try:
        ys = [K.convnext_mlp_fused(*args) for _ in range(3)]
        monkeypatch.setenv("OSP_MLP_SPLIT", "0")
        yf = K.convnext_mlp_fused(*args)
        torch.cuda.synchronize()
    finally:
        precision.set_precision("f32")
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])          # deterministic (no atomics in the pair reduction)
    scale = yf.abs().max().item()
    assert (ys[0] - yf).abs().max().item() <= 2e-6 * scale, ((ys[0] - yf).abs().max().item(), scale)
    ref = _reference(h, x, W1, W2, b1, b2, gamma, rowmask, rowscale)
    assert (ys[0].cpu().double() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item()


@pytest.mark.parametrize("C,I", [(384, 1152), (256, 1024)])
@pytest.mark.parametrize("hint", [None, "exact", "wrong"])
def test_masked_row_blocks_leave_early(C, I, hint):
    """Round 6: a padded batch (64 utterances x 772 frames, lengths 384 .. 772: a quarter of the rows masked, as in the synthesise
    benchmark).  Row blocks that are masked throughout are not computed -- zeros, as the mask would make them -- and the caller's count of
    live rows (kernels.live_rows) only picks the full / split workgroup mix: with no hint, the exact count or a wrong one the output
    is the restatement's."""
    from optispeech_amd import kernels as K, precision
    B, T = 64, 772
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(384, T + 1, (B,), generator=g); lens[0] = T
    M = B * T
    h, x, W1, W2, b1, b2, gamma, _, _ = _case(M, C, I, 31 * C + I, False)
    rowmask = (torch.arange(T)[None, :] < lens[:, None]).float().reshape(M)
    ref = _reference(h, x, W1, W2, b1, b2, gamma, rowmask, None)
    d = lambda t: None if t is None else t.to(DEV)
    hb = h.to(DEV).to(torch.bfloat16)
    W1p, W2p = torch.nn.Parameter(W1.to(DEV)), torch.nn.Parameter(W2.to(DEV))
    args = (hb, W1p, d(b1), W2p, d(b2), d(gamma), d(x), d(rowmask), None)
    precision.set_precision("bf16")
    try:
        if hint is None:
            y = K.convnext_mlp_fused(*args)
        else:
            with K.live_rows(M, int(lens.sum()) if hint == "exact" else 5000):
                y = K.convnext_mlp_fused(*args)
        y_nan = torch.full_like(y, float("nan"))
        torch.cuda.synchronize()
    finally:
        precision.set_precision("f32")
    y = y.cpu().double()
    scale = ref.abs().max().item()
    assert torch.isfinite(y).all() and (y[rowmask == 0] == 0).all()
    assert (y - ref).abs().max().item() <= 4e-3 * scale
    assert ((y - ref) ** 2).mean().sqrt().item() <= 3e-4 * scale


@pytest.mark.parametrize("M", [1, 5, 1023, 1024, 1025, 49344, 200001])
def test_row_order_is_a_stable_partition(M):
    from optispeech_amd import kernels as K
    g = torch.Generator().manual_seed(M)
    for frac in (0.0, 0.3, 1.0):
        mask = (torch.rand(M, generator=g) < frac).float().to(DEV)
        perm = torch.empty(M, dtype=torch.int32, device=DEV)
        K.call("osp_row_order", mask, perm, M)
        want = torch.argsort(mask.cpu() <= 0, stable=True).to(torch.int32)
        assert torch.equal(perm.cpu(), want)


def test_fused_mlp_is_what_the_no_grad_block_runs(monkeypatch):
    """ConvNeXtBlockFn under no_grad in performance mode = dwconv7+LN kernel + ONE MLP launch, equal to the autograd-capable path."""
    from optispeech_amd import kernels as K, ops, precision
    from optispeech_amd.model.modules import ConvNeXtBlock
    torch.manual_seed(0)
    blk = ConvNeXtBlock(384, 1152, layer_scale_init_value=0.3).to(DEV)
    with torch.no_grad():
        for w in (blk.dwconv_weight, blk.pwconv1_weight, blk.pwconv2_weight):
            w.normal_(0, 0.08)
    x = torch.randn(3, 217, 384, device=DEV)
    calls = []
    real = K.convnext_mlp_fused
    monkeypatch.setattr(K, "convnext_mlp_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    precision.set_precision("bf16")
    try:
        with torch.no_grad():
            y = blk(x)
        assert calls == [1] and ops._FUSED_MLP
        y2 = blk(x.clone().requires_grad_(True))
        assert calls == [1]                                   # the gradient-capable forward keeps the two launches (saves u, g, z)
        torch.cuda.synchronize()
    finally:
        precision.set_precision("f32")
    scale = y2.abs().max().item()
    assert (y - y2.detach()).abs().max().item() <= 4e-3 * scale


def test_padded_batch_walk_inside_a_replayed_decode_graph():
    """The live-row count is baked into a captured decode graph at its first use; the row order is recomputed inside the graph from the
    mask of the batch being replayed.  A second batch with the same padded shape and other lengths through the SAME graphs gives the
    eager result of that batch (the count only picks the workgroup mix: f32 summation order of the split mode, nothing else)."""
    from optispeech_amd import precision
    from optispeech_amd.config import ModelConfig, make_optispeech
    from optispeech_amd.values import InferenceInputs
    precision.set_precision("bf16")
    try:
        torch.manual_seed(0)
        m = make_optispeech(ModelConfig(), batch_size=32, pretraining_steps=0).to(DEV).eval()
        g = torch.Generator().manual_seed(3)
        n, Tt = 24, 96
        dur = torch.randint(4, 9, (n, Tt), generator=g)
        dur[0] = 8                                                                # sentence 0 is the longest in both batches: same y_max
        outs = {}
        for tag, lo in (("a", 70), ("b", 24)):
            xl = torch.randint(lo, Tt - 8, (n,), generator=g); xl[0] = Tt
            x = torch.randint(1, 159, (n, Tt), generator=g) * (torch.arange(Tt)[None] < xl[:, None])
            outs[tag] = InferenceInputs(clean_text="", x=x, x_lengths=xl, d_factor=1.0, p_factor=1.0, e_factor=1.0)
        res = {}
        for graph in (True, False):
            m.generator.graph_decode = graph
            for tag in ("a", "b", "a"):
                res[(graph, tag)] = m.synthesise(outs[tag], durations_override=dur)
        assert len(m.generator._decode_graphs) == 1                                # one capture served all three graph-mode calls
        for tag in ("a", "b"):
            og, oe = res[(True, tag)], res[(False, tag)]
            wg, we = torch.as_tensor(og.wav).double(), torch.as_tensor(oe.wav).double()
            assert torch.equal(torch.as_tensor(og.wav_lengths), torch.as_tensor(oe.wav_lengths))
            assert wg.shape == we.shape and torch.isfinite(wg).all()
            assert (wg - we).abs().max().item() <= 1e-4 * we.abs().max().item(), (tag, (wg - we).abs().max().item(), we.abs().max().item())
    finally:
        m.generator.graph_decode = False
        precision.set_precision("f32")
