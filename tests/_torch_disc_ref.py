"""Test-side torch (ATen conv2d) restatement of the MPD / MRD stacks, evaluated on a product module's own parameters
(reference: vocoder/wavenext/disc/_discriminators.py:63-97 and :165-194).  The product has no torch conv2d path; this is
what the f32 parity mode of the hand-written stacks is compared against."""
import torch
import torch.nn.functional as F


def _w(conv):
    v = conv.weight_v
    return v * (conv.weight_g / v.flatten(1).norm(dim=1).view(-1, 1, 1, 1))


def _conv(conv, x):
    return F.conv2d(x, _w(conv), conv.bias, conv.stride, conv.padding)


def disc_p(d, x):
    x = x.unsqueeze(1)
    b, c, t = x.shape
    if t % d.period != 0:
        n_pad = d.period - (t % d.period)
        x = F.pad(x, (0, n_pad), "reflect")
        t = t + n_pad
    x = x.view(b, c, t // d.period, d.period)
    fmap = []
    for i, conv in enumerate(d.convs):
        x = F.leaky_relu(_conv(conv, x), d.lrelu_slope)
        if i > 0:
            fmap.append(x)
    x = _conv(d.conv_post, x)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def disc_r(d, x):
    fmap = []
    x = d.spectrogram(x).unsqueeze(1)                 # |STFT| from the product's kernel: the stacks are what is compared
    for conv in d.convs:
        x = F.leaky_relu(_conv(conv, x), d.lrelu_slope)
        fmap.append(x)
    x = _conv(d.conv_post, x)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def multi(m, y, y_hat):
    """(real scores, generated scores, real fmaps, generated fmaps) like _Multi.forward in the discriminator phase."""
    one = disc_p if hasattr(m.discriminators[0], "period") else disc_r
    rs, gs, frs, fgs = [], [], [], []
    for d in m.discriminators:
        r, fr = one(d, y)
        g, fg = one(d, y_hat)
        rs.append(r); gs.append(g); frs.append(fr); fgs.append(fg)
    return rs, gs, frs, fgs
