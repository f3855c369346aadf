"""Oracle (test infrastructure): OptiSpeechGenerator.forward / .synthesise and the GAN training step.

Citations: optispeech/model/generator/__init__.py (G), optispeech/model/base_lightning_module.py (L).
Randomness (dropout, drop-path, segment starts) is injected explicitly so that the HIP path and
the reference can be driven with the same draws; ``None`` means rate 0 / eval mode.
"""
import math

import torch

from . import alignment as A
from . import disc as D
from . import losses as Ls
from . import nn_ops as N

LAMBDAS = dict(align=5.0, duration=1.0, pitch=1.0, energy=1.0)   # configs/model/generator/default.yaml:13-17


def generator_forward(P, batch, start_idx=None, rand01=None, segment_size=64, hop=256, pre="generator.",
                      lambdas=LAMBDAS, rng=None, keep=False):
    """OptiSpeechGenerator.forward G:72-192.

    batch: x (B,Tt) int64, x_lengths, mel (B,n_feats,Tm) [reference layout], mel_lengths, pitches (B,Tm),
    energies (B,Tm).  Either ``start_idx`` (B,) or ``rand01`` (B,) uniform draws fix the segment starts.
    ``rng``: optional dict of dropout / drop-path masks (see tests); None = all rates 0.
    Returns a dict with the reference's outputs plus named intermediates when keep=True.
    """
    rng = rng or {}
    x_tok, x_len, mel, mel_len = batch["x"], batch["x_lengths"], batch["mel"], batch["mel_lengths"]
    Tt, Tm = int(x_len.max()), int(mel_len.max())
    x_valid = N.length_mask(x_len, Tt)                                   # G:96-97
    m_valid = N.length_mask(mel_len, Tm)                                 # G:99-100
    x_pad, m_pad = ~x_valid, ~m_valid                                    # G:102-103

    x = N.text_embedding(x_tok, P, pre + "text_embedding.", drop_mask=rng.get("text_emb"))   # G:106
    enc = N.convnext_backbone(x, P, pre + "encoder.", x_pad, rng.get("encoder_dp"))          # G:109
    feats = mel.transpose(1, 2)                                                               # G:122
    log_p_attn = A.alignment_logprob(enc, feats, x_len, mel_len, x_pad, P, pre + "alignment_module.")   # G:120-126
    durations, bin_loss, paths = A.viterbi_decode(log_p_attn, x_len, mel_len)                # G:127
    d_hat = N.variance_predictor(enc.detach(), x_pad, P, pre + "duration_predictor.", rng.get("dur_drop"))  # G:128
    p_avg = A.average_by_duration(durations, batch["pitches"], x_len, mel_len)               # G:131
    e_avg = A.average_by_duration(durations, batch["energies"], x_len, mel_len)              # G:132
    p_hat = N.variance_predictor(enc, x_pad, P, pre + "pitch_predictor.predictor.", rng.get("pitch_drop"))   # G:135
    xp = N.variance_embed_add(enc, p_avg, x_pad, P, pre + "pitch_predictor.", rng.get("pitch_emb_drop"))
    e_hat = N.variance_predictor(xp, x_pad, P, pre + "energy_predictor.predictor.", rng.get("energy_drop"))  # G:136
    xe = N.variance_embed_add(xp, e_avg, x_pad, P, pre + "energy_predictor.", rng.get("energy_emb_drop"))
    y_up = A.gaussian_upsampling(xe, durations, m_valid, x_valid)                            # G:139-141
    dec = N.convnext_backbone(y_up, P, pre + "decoder.", m_pad, rng.get("decoder_dp"))       # G:144
    seg = min(segment_size, dec.shape[1])                                                    # G:147
    if start_idx is None:
        start_idx = A.segment_starts((mel_len - 4).float(), seg, rand01)                     # G:148-153
    segment = A.gather_segments(dec, start_idx, seg)                                         # G:149
    wav_hat = N.wavenext(segment.detach(), P, pre + "vocoder.", None, rng.get("vocoder_dp"))  # G:161
    d_loss, p_loss, e_loss = Ls.variance_losses(d_hat, p_hat, e_hat, durations, p_avg, e_avg, x_len)   # G:165-173
    fs_loss = Ls.forward_sum_loss(log_p_attn, x_len, mel_len)                                # G:174
    align_loss = fs_loss + bin_loss                                                          # G:175
    loss = (align_loss * lambdas["align"] + d_loss * lambdas["duration"] + p_loss * lambdas["pitch"]
            + e_loss * lambdas["energy"])                                                    # G:176-181
    out = dict(wav_hat=wav_hat, start_idx=start_idx, segment_size=seg, loss=loss, align_loss=align_loss,
               duration_loss=d_loss, pitch_loss=p_loss, energy_loss=e_loss)
    if keep:
        out.update(text_emb=x, enc=enc, log_p_attn=log_p_attn, durations=durations, paths=paths,
                   bin_loss=bin_loss, forwardsum_loss=fs_loss, d_hat=d_hat, p_hat=p_hat, e_hat=e_hat,
                   p_avg=p_avg, e_avg=e_avg, xp=xp, xe=xe, y_up=y_up, dec=dec, segment=segment)
    return out


@torch.no_grad()
def synthesise(P, x_tok, x_len, d_factor=1.0, p_factor=1.0, e_factor=1.0, hop=256, pre="generator.",
               durations_override=None, keep=False):
    """OptiSpeechGenerator.synthesise G:194-301 (single speaker / language)."""
    Tt = int(x_len.max())
    x_valid = N.length_mask(x_len, Tt)
    x_pad = ~x_valid
    x = N.text_embedding(x_tok, P, pre + "text_embedding.")                                  # G:229
    enc = N.convnext_backbone(x, P, pre + "encoder.", x_pad)                                 # G:232
    log_d = N.variance_predictor(enc, x_pad, P, pre + "duration_predictor.")
    durations = N.duration_infer(log_d, x_pad, d_factor)                                     # G:249
    if durations_override is not None:
        durations = durations_override.masked_fill(x_pad, 0)
    pitch = N.variance_predictor(enc, x_pad, P, pre + "pitch_predictor.predictor.") * p_factor   # G:252
    xp = N.variance_embed_add(enc, pitch, x_pad, P, pre + "pitch_predictor.")
    energy = N.variance_predictor(xp, x_pad, P, pre + "energy_predictor.predictor.") * e_factor  # G:254
    xe = N.variance_embed_add(xp, energy, x_pad, P, pre + "energy_predictor.")
    y_len = durations.sum(dim=1)                                                             # G:258
    y_valid = N.length_mask(y_len, int(y_len.max()))
    y_up = A.gaussian_upsampling(xe, durations.float(), y_valid, x_valid)                    # G:263-265
    dec = N.convnext_backbone(y_up, P, pre + "decoder.", ~y_valid)                           # G:268
    f0, _ = A.expand_by_duration(pitch[..., None], durations)                                # G:273-276
    wav = N.wavenext(dec, P, pre + "vocoder.", ~y_valid)                                     # G:277-281
    out = dict(wav=wav, wav_lengths=y_len * hop, durations=durations, pitch=pitch, energy=energy)
    if keep:
        out.update(enc=enc, xe=xe, y_up=y_up, dec=dec, f0=f0)
    return out


def ground_truth_segments(wav, start_idx, segment_size, hop=256):
    """_process_batch L:38-44: host-side slice of the ground-truth waveform. wav (B,Tw) float32."""
    return torch.stack([wav[i, int(s) * hop: int(s) * hop + segment_size * hop] for i, s in enumerate(start_idx)])


def cosine_warmup_lr(step, base_lr, warmup, total):
    """transformers.get_cosine_schedule_with_warmup lambda (num_cycles=0.5), as configured by
    configs/model/scheduler/cosine_with_warmup.yaml and L:58-67."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * prog)))


def training_step(P, batch, start_idx=None, rand01=None, fb=None, train_discriminator=True, clip=10.0,
                  with_mel=True, keep=False):
    """BaseLightningModule.training_step L:78-126 without the optimiser update: returns losses and the
    gradients of the G phase (w.r.t. generator.*) and of the D phase (w.r.t. discriminator.*).

    D phase sees ``wav_hat.detach()`` (the reference's cached non-detached wav_hat cannot
    be back-propagated twice).  P: dict name -> leaf tensor with requires_grad set by the caller.
    """
    gen_names = [k for k in P if k.startswith("generator.")]
    disc_names = [k for k in P if k.startswith("discriminator.")]
    out = generator_forward(P, batch, start_idx, rand01, keep=keep)
    wav = ground_truth_segments(batch["wav"], out["start_idx"], out["segment_size"])
    res = dict(out=out, wav=wav)
    loss_g = out["loss"]
    if train_discriminator:
        for k in disc_names:
            P[k].requires_grad_(False)                                      # toggle_optimizer L:93
        adv, logs = D.forward_gen(wav, out["wav_hat"], P, fb, with_mel=with_mel)          # L:142
        loss_g = loss_g + adv
        res.update(gen_adv_loss=adv, gen_logs=logs)
    gp = [P[k] for k in gen_names]
    grads = torch.autograd.grad(loss_g, gp, allow_unused=True)              # manual_backward L:99
    res["loss_g"] = loss_g.detach()
    res["grads_g"] = {k: g for k, g in zip(gen_names, grads)}
    if train_discriminator:
        for k in disc_names:
            P[k].requires_grad_(True)
        loss_d, dlogs = D.forward_disc(wav, out["wav_hat"].detach(), P)     # L:163-170
        dp = [P[k] for k in disc_names if P[k].is_floating_point() and P[k].requires_grad]
        dn = [k for k in disc_names if P[k].is_floating_point() and P[k].requires_grad]
        dgr = torch.autograd.grad(loss_d, dp, allow_unused=True)            # L:119
        res.update(loss_d=loss_d.detach(), disc_logs=dlogs, grads_d={k: g for k, g in zip(dn, dgr)})
    return res
