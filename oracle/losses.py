"""Oracle (test infrastructure): acoustic-model losses and STFT-family losses."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .nn_ops import length_mask


# --------------------------------------------------------------------------- A13
def variance_losses(d_hat, p_hat, e_hat, ds, ps, es, ilens, clip_val=1e-8):
    """FastSpeech2Loss.forward generator/loss.py:83-140 (use_masking branch, "l1" == SmoothL1 :77-78)
    + DurationPredictorLoss.forward :28-46.  All inputs (B,T)."""
    # Reference quirk that parity must reproduce: make_non_pad_mask returns (B,1,T) (utils/model.py:19-21);
    # masked_select of the (B,T,1) predictions against it broadcasts to (B,T,T) for durations (:115-116) and,
    # after unsqueeze(-1) -> (B,1,T,1), to (B,B,T,1) for pitch/energy (:117-120).  The "masked mean" is therefore
    #   duration: every position t < T of item b (padded ones included) counted L_b times;
    #   pitch/energy: every item's position t counted c_t = #{b': L_b' > t} times.
    # With all lengths == T both reduce to the plain mean.
    B, T = d_hat.shape
    valid = length_mask(ilens, T).float()                       # (B,T)
    L = valid.sum(dim=1)                                        # (B,)
    c = valid.sum(dim=0)                                        # (T,)
    d_err = (d_hat - torch.log(ds.float() + clip_val)) ** 2     # :43-44
    d_loss = (d_err * L[:, None]).sum() / (T * L.sum())
    p_loss = (F.smooth_l1_loss(p_hat, ps, reduction="none") * c[None, :]).sum() / (B * c.sum())
    e_loss = (F.smooth_l1_loss(e_hat, es, reduction="none") * c[None, :]).sum() / (B * c.sum())
    return d_loss, p_loss, e_loss


# --------------------------------------------------------------------------- A14
def forward_sum_loss(log_p_attn, ilens, olens, blank_prob=math.e ** -1):
    """ForwardSumLoss.forward generator/loss.py:150-194."""
    B = log_p_attn.shape[0]
    padded = F.pad(log_p_attn, (1, 0, 0, 0, 0, 0), value=math.log(blank_prob))    # :176
    loss = 0
    for b in range(B):
        N, T = int(ilens[b]), int(olens[b])
        target = torch.arange(1, N + 1)[None]                                      # :182
        cur = F.log_softmax(padded[b, :T, : N + 1][:, None, :], dim=-1)            # :183-186
        loss = loss + F.ctc_loss(cur, target, input_lengths=olens[b:b + 1], target_lengths=ilens[b:b + 1],
                                 zero_infinity=True)                               # :187-193
    return loss / B


# --------------------------------------------------------------------------- A15
def stft_mag(x, n_fft, hop, win, window, clamp=None):
    """|STFT| with center=True reflect padding. x (B,T) -> (B, frames, bins).

    clamp=1e-7: disc/loss.py:123-142 (`sqrt(clamp(re^2+im^2, 1e-7))`, transposed to (B,frames,bins)).
    clamp=None: plain `.abs()` as DiscriminatorR.spectrogram (_discriminators.py:196-216) / torchaudio.
    """
    s = torch.stft(x, n_fft, hop, win, window, center=True, pad_mode="reflect", return_complex=True)
    if clamp is None:
        return s.abs().transpose(1, 2)
    return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=clamp)).transpose(1, 2)


MRSTFT_RESOLUTIONS = ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240))       # disc/loss.py:150-153


def mr_stft_loss(x, y):
    """MultiResolutionSTFTLoss.forward disc/loss.py:169-194 (+ STFTLoss :211-228, SC :240-249, logmag :261-270).
    x = predicted, y = ground truth.  Returns (sc_loss, mag_loss)."""
    sc, mag = 0.0, 0.0
    for n_fft, hop, win in MRSTFT_RESOLUTIONS:
        w = torch.hann_window(win)
        xm, ym = stft_mag(x, n_fft, hop, win, w, 1e-7), stft_mag(y, n_fft, hop, win, w, 1e-7)
        sc = sc + torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
        mag = mag + F.l1_loss(torch.log(ym), torch.log(xm))
    return sc / len(MRSTFT_RESOLUTIONS), mag / len(MRSTFT_RESOLUTIONS)


def mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max):
    """torchaudio.functional.melscale_fbanks(mel_scale="htk", norm=None) restated from its documented
    definition (PARITY UNPINNED: torchaudio 2.5.1 is not installable in the build container).
    Returns (n_fft//2+1, n_mels) float32."""
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


def mel_spectrogram(x, fb, n_fft=1024, hop=256, win=1024):
    """torchaudio MelSpectrogram(power=1, center=True, hann periodic, norm=None, htk) call at
    disc/loss.py:94-107 -> (B, frames, n_mels)."""
    mag = stft_mag(x, n_fft, hop, win, torch.hann_window(win), None)
    return mag @ fb


def mel_l1_loss(y_hat, y, fb, clip_val=1e-7, **kw):
    """MelSpecReconstructionLoss.forward disc/loss.py:109-120 (safe_log = log(clip(x, 1e-7)), utils/model.py:168)."""
    a = torch.log(torch.clip(mel_spectrogram(y_hat, fb, **kw), min=clip_val))
    b = torch.log(torch.clip(mel_spectrogram(y, fb, **kw), min=clip_val))
    return F.l1_loss(b, a)


# --------------------------------------------------------------------------- A17 losses
def hinge_g(outs):
    """GeneratorLoss disc/loss.py:16-32 -> (sum, n)."""
    return sum(torch.mean(torch.clamp(1 - o, min=0)) for o in outs), len(outs)


def hinge_d(real, fake):
    """DiscriminatorLoss disc/loss.py:40-65 -> (sum, n)."""
    tot = 0
    for r, g in zip(real, fake):
        tot = tot + torch.mean(torch.clamp(1 - r, min=0)) + torch.mean(torch.clamp(1 + g, min=0))
    return tot, len(real)


def feature_matching(fr, fg):
    """FeatureMatchingLoss disc/loss.py:71-85."""
    tot = 0
    for dr, dg in zip(fr, fg):
        for a, b in zip(dr, dg):
            tot = tot + torch.mean(torch.abs(a - b))
    return tot
