#!/usr/bin/env python3
"""Phased 8-wave conv-GEMM (csrc/gemm_bf16_w8p.hip) against the lock-step kernel: bit-identity of the outputs (both accumulate
k-slabs in the same order on the same MFMA), error against torch's conv, and time per launch, in ONE process (OSP_GEMM_W8P is
read per call).  Shapes: the DiscriminatorP layers the 8-wave kernel serves (forward + fused-phase dgrad with the LeakyReLU'
epilogue), tile counts around one round of the chip, and plain GEMMs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import disc_ops as D, kernels as K, precision
precision.set_precision("bf16")
dev = "cuda"
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1,2").split(",")]
REPS = int(os.environ.get("REPS", "30"))
SCREEN = int(os.environ.get("SCREEN", "10"))


def timeit(f, reps=REPS):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


BASE_W8 = os.environ.get("OSP_GEMM_W8")                    # variant -1 = the 4-wave 128 x 128 kernel (OSP_GEMM_W8=0): timed, not compared


def set_variant(v):
    if v < 0:
        os.environ["OSP_GEMM_W8"] = "0"
    elif BASE_W8 is None:
        os.environ.pop("OSP_GEMM_W8", None)
    else:
        os.environ["OSP_GEMM_W8"] = BASE_W8
    os.environ["OSP_GEMM_W8P"] = str(max(v, 0))


def run_variants(label, fl, fn, ref=None):
    outs, ts = {}, {}
    for rnd in range(2):                                   # two interleaved rounds: the second one is reported (clock settled)
        for v in VARIANTS:
            set_variant(v)
            if rnd == 0:
                outs[v] = fn().clone()
            ts[v] = timeit(fn)
    cmp = [v for v in VARIANTS if v >= 0]
    base = outs[cmp[0]]
    msg = []
    bad = 0
    for v in cmp[1:]:                                      # race screen: SCREEN more launches of every phased variant, each compared bitwise
        set_variant(v)
        for _ in range(SCREEN):
            bad += int(not torch.equal(fn(), base))
    if bad:
        msg.append(f"RACE SCREEN: {bad} differing launches")
    for v in VARIANTS:
        same = v < 0 or torch.equal(outs[v], base)
        msg.append(f"v{v} {ts[v]:7.1f} us {fl / ts[v] / 1e6:5.0f} TF{'' if same else ' DIFFERS'}")
        if not same:
            d = (outs[v].float() - base.float()).abs()
            msg.append(f"(max diff {d.max().item():.3e}, {int((d > 0).sum())} elements)")
    if ref is not None:
        err = (base.float() - ref).abs().max().item() / ref.abs().max().item()
        msg.append(f"err vs torch {err:.1e}")
        assert err < 2e-2, err
    print(f"{label}: " + " | ".join(msg), flush=True)
    return bad == 0 and all(torch.equal(outs[v], base) for v in cmp)


if os.environ.get("ONE"):                                   # one shape, N launches of the variant in OSP_GEMM_W8P: what tools/probes/w8p_pmc.sh profiles
    U, T, cin, n = int(os.environ.get("U", "160")), 102, 1024, 1024
    a = torch.randn(U * T, cin, device=dev).bfloat16(); w = (torch.randn(n, 5, cin, device=dev) * 0.03).bfloat16()
    for _ in range(int(os.environ.get("N", "10"))):
        K.conv_gemm_bf16(a, w, n, M=U * T, Trows=T, Tin=T, cin=cin, taps=5, a_step=1, a_off=-2, out_bf16=True)
    torch.cuda.synchronize()
    sys.exit(0)
ok = True
# 1. plain 5-tap convs at tile counts around one round of the chip (the shapes of round 5's fill sweep, profiles/r05_w8_fill_sweep.txt)
for (U, T, cin, n, st) in [(128, 102, 1024, 1024, 1), (160, 102, 1024, 1024, 1), (256, 102, 1024, 1024, 1), (128, 304, 512, 1024, 3),
                           (704, 19, 1024, 1024, 1), (64, 102, 1024, 1024, 1), (131, 97, 1024, 768, 1)]:
    Tout = (T + 4 - 5) // st + 1
    M = U * Tout
    a = torch.randn(U * T, cin, device=dev).bfloat16(); w = (torch.randn(n, 5, cin, device=dev) * 0.03).bfloat16()
    import torch.nn.functional as F
    ref = F.conv1d(a.float().view(U, T, cin).transpose(1, 2), w.float().permute(0, 2, 1).contiguous(), stride=st, padding=2).transpose(1, 2).reshape(M, n)
    ok &= run_variants(f"conv U={U} T={T} {n}<-{cin} s{st} M={M} tiles={-(-M // 256) * -(-n // 256)}", 2.0 * M * n * cin * 5,
                       lambda: K.conv_gemm_bf16(a, w, n, M=M, Trows=Tout, Tin=T, cin=cin, taps=5, a_step=st, a_off=-2, out_bf16=True), ref)
    del ref
# 2. the DiscriminatorP layers of the step: forward with LeakyReLU, fused-phase dgrad with LeakyReLU'
tot = {v: 0.0 for v in VARIANTS}
for p in (2, 3, 5, 7, 11):
    Ub = 64 * p
    T0 = -(-16384 // p)
    Ws = [T0]
    for _ in range(4):
        Ws.append((Ws[-1] + 4 - 5) // 3 + 1)
    for (cin, cout, li, s) in ((128, 512, 2, 3), (512, 1024, 3, 3), (1024, 1024, 4, 1)):      # (128 -> 512: 8-wave only under OSP_GEMM_W8=1)
        W = Ws[li]
        Wo = (W + 4 - 5) // s + 1
        x = torch.randn(Ub, 1, W, cin, device=dev).bfloat16()
        w = (torch.randn(cout, 1, 5, cin, device=dev) * 0.02).bfloat16()
        wt = w.permute(3, 1, 2, 0).contiguous()
        bias = torch.randn(cout, device=dev) * 0.1
        dy = torch.randn(Ub, 1, Wo, cout, device=dev).bfloat16()
        fl = 2.0 * Ub * Wo * 5 * cin * cout
        ok &= run_variants(f"fwd  p={p:2d} {cin}->{cout} s{s} M={Ub * Wo}", fl, lambda: D.conv2d_fwd(x, w, bias, 1, 5, 1, s, 0, 2, 0.1, True))
        ok &= run_variants(f"dgrd p={p:2d} {cin}->{cout} s{s} M={Ub * W}", fl,
                           lambda: D.conv2d_dgrad(dy, wt, 1, W, 1, 5, 1, s, 0, 2, lrelu_y=x, slope=0.1, out_bf16=True))
# 3. plain GEMMs (one tap)
for (M, N, Kd) in [(16384, 4096, 4096), (8192, 8192, 8192), (4096, 4096, 4096), (13056, 1024, 5120)]:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w = torch.randn(N, Kd, device=dev).bfloat16()
    ok &= run_variants(f"gemm M={M} N={N} K={Kd}", 2.0 * M * N * Kd, lambda: K.conv_gemm_bf16(a, w, N, M=M, Trows=M, Tin=M, cin=Kd, out_bf16=True))
    tt = timeit(lambda: torch.matmul(a, w.t()))
    print(f"     hipBLASLt {tt:8.1f} us {2.0 * M * N * Kd / tt / 1e6:6.0f} TF", flush=True)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
