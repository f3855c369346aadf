#!/usr/bin/env python3
"""Headline benchmark: full OptiSpeech ConvNeXt GAN training step on synthetic LJSpeech-shaped batches.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = BaseLightningModule.training_step in the post-pre-training regime: generator forward, adversarial
losses through MPD/MRD, G backward, clip + AdamW(G), discriminator forward/backward, clip + AdamW(D); train mode
(dropout / drop-path active); --precision bf16 (default, BASELINE config[1]) or f32 (exact parity mode).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON
line; `value` is the whole-job aggregate (mel-frames/s over all ranks, weak scaling: 32 utterances per GPU).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, T_TEXT, T_MEL = 32, 128, 800
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 MFMA peak (same guide; AMD's 5 PF figure is 2:1 sparse)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-am-only", action="store_true", help="skip the acoustic-model-only (pre-training regime) step timing")
    ap.add_argument("--no-infer", action="store_true", help="skip the synthesise() RTF measurement (secondary metric)")
    ap.add_argument("--cpu-batch", type=int, default=4, help="utterances in the bounded CPU-baseline sample")
    ap.add_argument("--ragged", action="store_true", help="ragged lengths (BASELINE.md section 3 variant)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="serial schedule: do not issue the discriminator phase from its own stream (OptiSpeech.pipeline_steps)")
    ap.add_argument("--backbone", choices=["convnext", "transformer"], default="convnext",
                    help="transformer = BASELINE configs[4] encoder/decoder (secondary datapoint; the headline is convnext)")
    ap.add_argument("--precision", choices=["bf16", "f32"], default=os.environ.get("OSP_PRECISION", "bf16"),
                    help="bf16 = BASELINE config[1] (bf16 MFMA operands, f32 accumulate, f32 master weights); "
                         "f32 = exact-f32 parity mode")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of selected C-ABI launches on the launch stream (torch's current stream) inside the timed region,
    so `roofline.achieved` = algorithmic flops of those launches / their measured duration."""

    def __init__(self, select):
        self.select, self.events, self.enabled = select, [], False

    def install(self):
        from optispeech_amd import _lib
        lib = _lib.lib()
        orig = lib.call
        timer = self

        def call(name, *args):
            fl = timer.select(name, args) if timer.enabled else None
            if fl:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig(name, *args)
                e1.record()
                timer.events.append((fl, e0, e1))
            else:
                orig(name, *args)
        lib.call = call

    def summary(self):
        if not self.events:
            return None, None, 0
        ms = sum(a.elapsed_time(b) for _, a, b in self.events)
        return sum(f for f, _, _ in self.events), ms, len(self.events)


def cpu_baseline(nb):
    """Oracle (CPU port of the reference step) on a bounded sample: `nb` utterances of the same shape, one
    untimed + one timed full GAN step incl. torch.optim.AdamW updates.  Returns (frames/s, threads, note)."""
    from oracle import generator as OG
    from oracle import losses as OL
    from oracle import schema as S
    from optispeech_amd.config import ModelConfig, synthetic_batch
    threads = torch.get_num_threads()
    P = S.make_weights(S.generator_schema(S.Cfg()), 1)
    P.update(S.make_weights(S.discriminator_schema(), 2))
    for v in P.values():
        v.requires_grad_(True)
    fb = OL.mel_filterbank(22050, 1024, 100, 80, 8000)
    gp = [v for k, v in P.items() if k.startswith("generator.")]
    dp_ = [v for k, v in P.items() if k.startswith("discriminator.")]
    og = torch.optim.AdamW(gp, lr=2e-4, betas=(0.8, 0.99), weight_decay=1e-2)
    od = torch.optim.AdamW(dp_, lr=2e-4, betas=(0.8, 0.99), weight_decay=1e-2)
    batch = synthetic_batch(nb, T_TEXT, T_MEL, ModelConfig(), seed=3)
    rand01 = torch.rand(nb)

    def step():
        res = OG.training_step(P, batch, rand01=rand01, fb=fb, train_discriminator=True, with_mel=True)
        for k, g in res["grads_g"].items():
            P[k].grad = g
        torch.nn.utils.clip_grad_norm_([p for p in gp if p.grad is not None], 10.0)
        og.step()
        for k, g in res["grads_d"].items():
            P[k].grad = g
        torch.nn.utils.clip_grad_norm_(dp_, 10.0)
        od.step()

    step()
    t0 = time.perf_counter()
    step()
    dt = time.perf_counter() - t0
    return nb * T_MEL / dt, threads, f"{nb} utterances x (T_text={T_TEXT}, T_mel={T_MEL}), 1 warm + 1 timed GAN step, {dt:.1f}s"


def synthesise_rtf(model, dev, n_sent=64, seed=7):
    """BASELINE config[4]: synthesise() on 64 batched sentences; durations overridden to U{4..8} frames/phoneme because
    random-init weights predict degenerate durations.  RTF as the reference defines it
    (generator/__init__.py:285-288): (t_acoustic + t_vocoder) / (padded wav length / sample_rate)."""
    from optispeech_amd.values import InferenceInputs
    g = torch.Generator().manual_seed(seed)
    x_len = torch.randint(64, 129, (n_sent,), generator=g)
    x_len[0] = 128
    x = torch.randint(1, 159, (n_sent, 128), generator=g) * (torch.arange(128)[None] < x_len[:, None])
    dur = torch.randint(4, 9, (n_sent, 128), generator=g)
    inputs = InferenceInputs(clean_text="", x=x, x_lengths=x_len, d_factor=1.0, p_factor=1.0, e_factor=1.0)
    model.eval()
    outs = [model.synthesise(inputs, durations_override=dur) for _ in range(3)]
    model.train()
    o = outs[-1]
    audio_s = float(o.wav_lengths.sum()) / model.sample_rate
    return {"rtf": o.rtf, "am_rtf": o.am_rtf, "v_rtf": o.v_rtf, "latency_ms": o.latency, "sentences": n_sent,
            "padded_audio_s": o.wav.shape[-1] / model.sample_rate, "total_audio_s": audio_s,
            "aggregate_audio_s_per_s": audio_s / (o.latency * 1e-3)}


def main():
    a = parse()
    from optispeech_amd import dp, precision, rng
    precision.set_precision(a.precision)
    world, rank, local = dp.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP extension has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch

    torch.manual_seed(1234)                                   # configs/train.yaml:53; same init on every rank
    rng.manual_seed(1234, rank)
    cfg = ModelConfig(backbone=a.backbone)
    model = make_optispeech(cfg, batch_size=B, pretraining_steps=0).to(dev).train()
    batch = synthetic_batch(B, T_TEXT, T_MEL, cfg, seed=1234 + rank, ragged=a.ragged, device=dev)
    model.optimizers()
    # production schedule: the discriminator phase runs from a second calling stream, so step n+1's generator forward
    # overlaps step n's discriminator backward (everything is drained by the synchronize() that closes the timed region)
    model.pipeline_steps = not a.no_pipeline

    # dominant hand-written kernel by time: conv_gemm_bf16_glds_kernel (csrc/gemm_bf16.hip), i.e. every conv-GEMM launch
    # whose operands are bf16 in HBM with Cin % 64 == 0 (all large MPD / MRD forward and dgrad GEMMs).  Algorithmic
    # flops per launch = 2 * M * taps * Cin * N; the launches are timed with HIP events on the launch
    # stream, so sum(flops) / sum(time) is comparable with the kernel's average in the rocprofv3 summary under profiles/.
    if a.precision == "bf16":
        def select(name, args):
            # exactly the launches conv_gemm_bf16_impl routes to conv_gemm_bf16_glds_kernel: bf16 A and B, k-contiguous B,
            # Cin % 64 == 0, N > 64 and at least 160 tiles of 128x128 (csrc/gemm_bf16.hip, tile selection)
            if name == "osp_conv2d_gemm_bf16" and args[1] == 1 and args[18] == 1 and args[22] == 1 and args[8] % 64 == 0:
                M_, N_ = args[3], args[23]
                if N_ > 64 and -(-M_ // 128) * -(-N_ // 128) >= 160:
                    return 2.0 * M_ * args[9] * args[8] * N_                          # M, taps, Cin, N
            # fused-phase dgrad (osp_conv2d_dgrad_bf16): the same kernel with blockIdx.z = output phase
            if name == "osp_conv2d_dgrad_bf16" and args[1] == 1 and args[3] == 1:
                U, H, W, Cin, Cout, KH, KW, sh, sw, ph, pw = (args[6], args[7], args[8], args[11], args[12], args[13], args[14],
                                                              args[15], args[16], args[17], args[18])
                if Cout % 64 == 0 and Cin > 64 and Cout > 1:
                    fl, mmax, nph = 0.0, 0, 0
                    for rh in range(sh):
                        for rw in range(sw):
                            qh, qw = (H - rh + sh - 1) // sh, (W - rw + sw - 1) // sw
                            if qh <= 0 or qw <= 0:
                                continue
                            n_h = (KH - (rh + ph) % sh + sh - 1) // sh
                            n_w = (KW - (rw + pw) % sw + sw - 1) // sw
                            fl += 2.0 * U * qh * qw * n_h * n_w * Cout * Cin
                            mmax, nph = max(mmax, U * qh * qw), nph + 1
                    if -(-mmax // 128) * -(-Cin // 128) * nph >= 160:
                        return fl
            return None
        roof_kernel, roof_peak = "conv_gemm_bf16_glds_kernel (MPD conv-GEMM forward + fused-phase dgrad launches, N >= 128)", PEAK_BF16_MFMA_TFLOPS
    else:
        M = B * T_MEL

        def select(name, args):
            if name == "osp_conv_gemm_f32" and args[2] == M and args[4] * args[12] == 256 * 1024 and args[5] == 1:
                return 2.0 * M * 256 * 1024
            return None
        roof_kernel, roof_peak = "conv_gemm_f32 (decoder pwconv1/pwconv2, M=25600, 256<->1024)", PEAK_F32_MFMA_TFLOPS
    timer = KernelTimer(select)
    timer.install()

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # set-up, not warm-up: two steps so that one-off costs (lazy kernel-attribute calls, stream creation, caching-allocator
    # growth, weight packs) are paid before the W warm-up steps the caller asked for, whatever W is
    for i in range(2):
        model.training_step(batch, i)
    for i in range(a.warmup):
        model.training_step(batch, 2 + i)
    sync()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(a.steps):
        model.training_step(batch, a.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    logs = model.fetch_logs()
    # Second, untimed look at the same kernel with the sub-discriminator streams switched off: inside the timed region the
    # eight discriminators run concurrently (optispeech_amd/model/discriminator.py), so a launch's event-to-event duration
    # there includes whatever shared the GPU with it; serialised launches give the kernel's own efficiency.
    iso = None
    if a.precision == "bf16":
        from optispeech_amd.model import discriminator as _disc
        if _disc._DISC_STREAMS:
            timed_events, timer.events = timer.events, []
            _disc._DISC_STREAMS = False
            timer.enabled = True
            for i in range(3):
                model.training_step(batch, a.warmup + a.steps + i)
            sync()
            timer.enabled = False
            _disc._DISC_STREAMS = True
            iso = timer.summary()
            timer.events = timed_events
    # secondary figure: the acoustic-model-only step of the first `pretraining_steps` steps (no adversarial
    # losses, no discriminator phase: base_lightning_module.py:88,108-110,149-150); same batch, 3 warm-up + 10 timed steps
    am_only = None
    if not a.no_am_only:
        keep, model.train_args.pretraining_steps = model.train_args.pretraining_steps, 1 << 60
        n0 = a.warmup + a.steps + 3
        for i in range(3):
            model.training_step(batch, n0 + i)
        sync()
        t1 = time.perf_counter()
        for i in range(10):
            model.training_step(batch, n0 + 3 + i)
        sync()
        am_dt = (time.perf_counter() - t1) / 10
        model.train_args.pretraining_steps = keep
        am_only = {"ms_per_step": am_dt * 1e3, "mel_frames_per_s": world * B * T_MEL / am_dt, "steps": 10,
                   "note": "pre-training regime (global_step < pretraining_steps): acoustic-model losses only, per-rank wall time of rank 0"}
    # secondary figure: the discriminator phase re-using the forward that the generator phase of the same step ran on the same
    # waves with the same (not yet updated) discriminator weights (OptiSpeech.replay_disc_forward; bit-identical values, the
    # reference evaluates them twice).  Not the headline: `value` above recomputes that forward, as the reference does.
    replay = None
    if not a.no_am_only and a.precision == "bf16":
        keep_r, model.replay_disc_forward = model.replay_disc_forward, True
        n1 = a.warmup + a.steps + 20
        for i in range(3):
            model.training_step(batch, n1 + i)
        sync()
        t2 = time.perf_counter()
        for i in range(10):
            model.training_step(batch, n1 + 3 + i)
        sync()
        r_dt = (time.perf_counter() - t2) / 10
        model.replay_disc_forward = keep_r
        replay = {"ms_per_step": r_dt * 1e3, "mel_frames_per_s": world * B * T_MEL / r_dt, "steps": 10,
                  "note": "OSP_DISC_REPLAY=1: discriminator-phase forward taken from the generator phase's recorded activations (same step, same weights, same waves)"}
    ms_per_step = dt / a.steps * 1e3
    value = world * B * T_MEL / (dt / a.steps)

    if rank == 0:
        flops, kms, nlaunch = timer.summary()
        roof = {"bound": "mfma", "kernel": roof_kernel,
                "achieved": (flops / (kms * 1e-3) / 1e12) if kms else None, "peak": roof_peak,
                "unit": "TFLOP/s", "traffic": None, "launches_timed": nlaunch,
                "avg_launch_us": (kms / nlaunch * 1e3) if nlaunch else None}
        roof["frac"] = (roof["achieved"] / roof["peak"]) if roof["achieved"] else None
        if iso and iso[1]:
            roof["concurrency"] = "timed region: sub-discriminators on 8 HIP streams (launch durations include co-scheduled kernels)"
            roof["isolated"] = {"achieved": iso[0] / (iso[1] * 1e-3) / 1e12, "frac": iso[0] / (iso[1] * 1e-3) / 1e12 / roof_peak,
                                "avg_launch_us": iso[1] / iso[2] * 1e3, "launches_timed": iso[2],
                                "note": "same kernel, 3 extra untimed steps with the launches serialised on one stream"}
        # HBM-side bytes per launch come from a separate rocprofv3 --pmc pass (counters cannot be read inside the timed
        # run); the committed summary of that pass is reported here when it matches the measured configuration
        pmc = os.path.join(ROOT, "profiles", "r01h_pmc_glds.json")
        if a.precision == "bf16" and not a.ragged and os.path.exists(pmc):
            with open(pmc) as fh:
                pj = json.load(fh)
            roof["traffic"] = pj["traffic_bytes_per_launch"]
            roof["traffic_unit"] = "bytes/launch (TCC_EA0 read x 128 B + write x 64 B, " + os.path.basename(pmc) + ")"
            roof["algorithmic_flop_per_launch"] = flops / nlaunch if nlaunch else None
        cpu = None
        if not a.no_cpu_baseline:
            v, threads, note = cpu_baseline(a.cpu_batch)
            cpu = {"value": v, "unit": "mel-frames/s", "cores": threads, "kind": "port", "sample": note}
        out = {"metric": "mel-frames/sec/GPU (train step) + RTF (synthesize), ConvNeXt@22.05kHz, 1/2/4/8 MI355X",
               "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.precision, "data": "synthetic",
               "config": {"workload": ("configs[4]: Transformer backbone" if a.backbone == "transformer" else "configs[1]: ConvNeXt backbone") + ", synthetic LJSpeech-shaped batch=32 per GPU "
                                      "(T_text=128, T_mel=800, 22.05 kHz), full GAN training step "
                                      "(G phase + D phase + 2x AdamW), train mode",
                          "global_batch": B * world, "T_text": T_TEXT, "T_mel": T_MEL, "parallelism": f"dp{world}", "schedule": "serial" if a.no_pipeline else "pipelined (pipeline_steps)",
                          "lengths": "ragged" if a.ragged else "fixed"},
               "per_gpu": value / world, "roofline": roof, "cpu_baseline": cpu,
               "am_only_step": am_only, "replay_disc_forward_step": replay,
               "synthesise": None if a.no_infer else synthesise_rtf(model, dev),
               "final_losses": {k: round(v, 5) for k, v in logs.items() if k.startswith("total_loss/")}}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
