"""Oracle (test infrastructure): VocosDiscriminator = MPD + MRD + GAN losses.

Citations: optispeech/model/vocoder/wavenext/disc/{__init__,_discriminators,loss}.py
"""
import torch
import torch.nn.functional as F

from . import losses

MPD_PERIODS = (2, 3, 5, 7, 11)                                                      # _discriminators.py:19
MRD_RESOLUTIONS = ((1024, 256, 1024), (2048, 512, 2048), (512, 128, 512))         # _discriminators.py:103
# (kernel, stride, padding) per conv
MRD_CONVS = (((7, 5), (2, 2), (3, 2)), ((5, 3), (2, 1), (2, 1)), ((5, 3), (2, 2), (2, 1)),
             ((3, 3), (2, 1), (1, 1)), ((3, 3), (2, 2), (1, 1)))                   # _discriminators.py:154-160
MPD_STRIDES = (3, 3, 3, 3, 1)                                                       # _discriminators.py:53-57


def wn_weight(P, pre):
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v|| with the norm over all dims but 0."""
    v, g = P[pre + "weight_v"], P[pre + "weight_g"]
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def disc_p(x, period, P, pre, slope=0.1):
    """DiscriminatorP.forward _discriminators.py:62-97. x (B,T)."""
    x = x[:, None, :]
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t = t + n_pad
    x = x.view(b, c, t // period, period)
    fmap = []
    for i, s in enumerate(MPD_STRIDES):
        x = F.conv2d(x, wn_weight(P, pre + f"convs.{i}."), P[pre + f"convs.{i}.bias"], stride=(s, 1), padding=(2, 0))
        x = F.leaky_relu(x, slope)
        if i > 0:
            fmap.append(x)
    x = F.conv2d(x, wn_weight(P, pre + "conv_post."), P[pre + "conv_post.bias"], stride=1, padding=(1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def disc_r(x, resolution, P, pre, slope=0.1):
    """DiscriminatorR.forward/.spectrogram _discriminators.py:164-216. x (B,T)."""
    n_fft, hop, win = resolution
    spec = losses.stft_mag(x, n_fft, hop, win, torch.ones(n_fft), None).transpose(1, 2)   # (B,freq,frames)
    x = spec[:, None]
    fmap = []
    for i, (k, s, p) in enumerate(MRD_CONVS):
        x = F.conv2d(x, wn_weight(P, pre + f"convs.{i}."), P[pre + f"convs.{i}.bias"], stride=s, padding=p)
        x = F.leaky_relu(x, slope)
        fmap.append(x)
    x = F.conv2d(x, wn_weight(P, pre + "conv_post."), P[pre + "conv_post.bias"], padding=(1, 1))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd(y, y_hat, P, pre):
    """MultiPeriodDiscriminator.forward _discriminators.py:23-38."""
    rs, gs, frs, fgs = [], [], [], []
    for i, p in enumerate(MPD_PERIODS):
        r, fr = disc_p(y, p, P, pre + f"discriminators.{i}.")
        g, fg = disc_p(y_hat, p, P, pre + f"discriminators.{i}.")
        rs.append(r); gs.append(g); frs.append(fr); fgs.append(fg)
    return rs, gs, frs, fgs


def mrd(y, y_hat, P, pre):
    """MultiResolutionDiscriminator.forward _discriminators.py:115-136."""
    rs, gs, frs, fgs = [], [], [], []
    for i, r_ in enumerate(MRD_RESOLUTIONS):
        r, fr = disc_r(y, r_, P, pre + f"discriminators.{i}.")
        g, fg = disc_r(y_hat, r_, P, pre + f"discriminators.{i}.")
        rs.append(r); gs.append(g); frs.append(fr); fgs.append(fg)
    return rs, gs, frs, fgs


def forward_disc(wav, wav_hat, P, pre="discriminator.", lambda_mrd=1.0):
    """VocosDiscriminator.forward_disc disc/__init__.py:44-61."""
    r_mp, g_mp, _, _ = mpd(wav, wav_hat, P, pre + "multiperioddisc.")
    r_mr, g_mr, _, _ = mrd(wav, wav_hat, P, pre + "multiresddisc.")
    l_mp, n_mp = losses.hinge_d(r_mp, g_mp)
    l_mr, n_mr = losses.hinge_d(r_mr, g_mr)
    l_mp, l_mr = l_mp / n_mp, l_mr / n_mr
    return l_mp + l_mr * lambda_mrd, {"loss_mp": l_mp, "loss_mrd": l_mr}


def forward_gen(wav, wav_hat, P, fb, pre="discriminator.", lambda_mrd=1.0, lambda_mel=45.0, lambda_mr_stft=2.5,
                with_mel=True):
    """VocosDiscriminator.forward_gen disc/__init__.py:63-96 (+ _get_mel_loss/_get_mr_stft_loss :105-111)."""
    _, g_mp, fr_mp, fg_mp = mpd(wav, wav_hat, P, pre + "multiperioddisc.")
    _, g_mr, fr_mr, fg_mr = mrd(wav, wav_hat, P, pre + "multiresddisc.")
    l_g_mp, n1 = losses.hinge_g(g_mp)
    l_g_mr, n2 = losses.hinge_g(g_mr)
    l_g_mp, l_g_mr = l_g_mp / n1, l_g_mr / n2
    fm_mp = losses.feature_matching(fr_mp, fg_mp) / len(fr_mp)
    fm_mr = losses.feature_matching(fr_mr, fg_mr) / len(fr_mr)
    mel = losses.mel_l1_loss(wav_hat, wav, fb) * lambda_mel if with_mel else torch.zeros(())
    sc, mag = losses.mr_stft_loss(wav_hat, wav)
    mr = (sc + mag) * lambda_mr_stft
    loss = l_g_mp + l_g_mr * lambda_mrd + fm_mp + fm_mr * lambda_mrd + mel + mr
    return loss, {"loss_gen_mp": l_g_mp, "loss_gen_mrd": l_g_mr, "loss_fm_mp": fm_mp, "loss_fm_mrd": fm_mr,
                  "mel_loss": mel, "mr_stft_loss": mr, "sc": sc, "mag": mag}
