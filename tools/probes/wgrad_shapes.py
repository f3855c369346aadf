#!/usr/bin/env python3
"""Weight-gradient launches of one B = 32 training step, shape by shape, 20 back-to-back launches each between HIP events
(round 5: the narrow-channel family).  Prints us / launch, TFLOP/s and the share of the dense bf16 peak."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optispeech_amd import kernels as K
from optispeech_amd import disc_ops as D

dev = "cuda"
REP = int(os.environ.get("REP", "20"))
ONLY = os.environ.get("ONLY", "")


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e3          # us


def report(name, us, flop):
    print(f"{name:58s} {us:8.1f} us  {flop / us / 1e6:7.1f} TF  {flop / us / 1e6 / 2500 * 100:5.1f} % of peak", flush=True)


def bf(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(torch.bfloat16)


def mrd(tag, U, H, W, KH, KW, sh, sw, ph, pw, C=64):
    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    x, dy = bf(U, H, W, C), bf(U, Ho, Wo, C)
    dw = torch.zeros(C, KH, KW, C, device=dev)
    db = None if os.environ.get("NOBIAS", "0") == "1" else torch.zeros(C, device=dev)
    M = U * Ho * Wo

    def f():
        K.conv2d_wgrad_bf16(dy.view(M, C), x.view(U * H * W, C), dw, db, M=M, Trows=Ho * Wo, Wrows=Wo, Hin=H, Win=W, n=C, cin=C,
                            taps=KH * KW, KW=KW, pad_h=ph, pad_w=pw, step_h=sh, step_w=sw)
    report(f"MRD {tag} M={M} 64<-64 ({KH},{KW}) s({sh},{sw}) Wo={Wo}", timeit(f), 2.0 * M * C * C * KH * KW)


def mpd(tag, U, Tin, cin, cout, st=3):
    Tout = (Tin + 4 - 5) // st + 1
    x, dy = bf(U, Tin, cin), bf(U, Tout, cout)
    dw = torch.zeros(cout, 5, cin, device=dev)
    db = None if os.environ.get("NOBIAS", "0") == "1" else torch.zeros(cout, device=dev)
    M = U * Tout

    def f():
        K.conv_wgrad_bf16(dy.view(M, cout), x.view(U * Tin, cin), dw, db, M=M, Trows=Tout, Tin=Tin, n=cout, cin=cin, taps=5, pad=2,
                          x_step=st)
    report(f"MPD {tag} M={M} {cout}<-{cin} k5 s{st}", timeit(f), 2.0 * M * cout * cin * 5)


def pw(tag, M, n, cin, taps=1, f32=False, T=None):
    x = torch.randn(M, cin, device=dev) if f32 else bf(M, cin)
    dy = torch.randn(M, n, device=dev) if f32 else bf(M, n)
    dw = torch.zeros(n, taps, cin, device=dev)
    db = None if os.environ.get("NOBIAS", "0") == "1" else torch.zeros(n, device=dev)
    T = M if T is None else T

    def f():
        K.conv_wgrad_bf16(dy, x, dw, db, M=M, Trows=T, Tin=T, n=n, cin=cin, taps=taps, pad=taps // 2)
    report(f"GEN {tag} M={M} {n}<-{cin} k{taps} {'f32' if f32 else 'bf16'}", timeit(f), 2.0 * M * n * cin * taps)


cases = {
    # DiscriminatorR, discriminator phase (batch 64 = real + generated), resolution 2048 / 1024 / 512: layers 2..5
    "mrd": lambda: [mrd("2048 L2", 64, 17, 513, 3, 5, 1, 2, 1, 2), mrd("2048 L3", 64, 17, 257, 3, 5, 2, 2, 1, 2),
                    mrd("2048 L4", 64, 9, 129, 3, 3, 1, 2, 1, 1), mrd("2048 L5", 64, 9, 65, 3, 3, 2, 2, 1, 1),
                    mrd("512 L2", 64, 65, 129, 3, 5, 1, 2, 1, 2), mrd("512 L3", 64, 65, 65, 3, 5, 2, 2, 1, 2),
                    mrd("512 L4", 64, 33, 33, 3, 3, 1, 2, 1, 1), mrd("512 L5", 64, 33, 17, 3, 3, 2, 2, 1, 1)],
    # DiscriminatorP period 2 / 11, layers 2..5 (batch 64)
    "mpd": lambda: [mpd("p2 L2", 128, 2731, 32, 128), mpd("p2 L3", 128, 911, 128, 512), mpd("p2 L4", 128, 304, 512, 1024),
                    mpd("p2 L5", 128, 102, 1024, 1024, 1), mpd("p11 L2", 704, 497, 32, 128), mpd("p11 L3", 704, 166, 128, 512),
                    mpd("p11 L4", 704, 56, 512, 1024), mpd("p11 L5", 704, 19, 1024, 1024, 1)],
    # generator: ConvNeXt pointwise pairs (vocoder 32 x 64 frames, encoder 32 x 128 tokens), text-side convs
    "gen": lambda: [pw("voc pw2", 2048, 384, 1152), pw("voc pw1", 2048, 1152, 384), pw("enc pw2", 4096, 256, 1024),
                    pw("enc pw1", 4096, 1024, 256), pw("pred k5", 4096, 256, 256, 5, f32=True, T=128),
                    pw("pred k3", 4096, 384, 384, 3, f32=True, T=128), pw("voc embed k7", 2048, 384, 256, 7, f32=True, T=64),
                    pw("align k3", 25600, 256, 256, 3, f32=True, T=800), pw("head", 2048, 1088, 384, f32=True)],
}
for k, fn in cases.items():
    if ONLY and k not in ONLY.split(","):
        continue
    fn()
