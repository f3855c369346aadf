"""STFT-family operators and losses (reference rows A15 / A15b): magnitudes come from the in-LDS FFT kernels
(csrc/stft.hip); module classes mirror optispeech/model/vocoder/wavenext/disc/loss.py and keep its buffer names."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import kernels as K
from ._lib import call

_TW = {}


def _publish(t):
    """A table filled on the CURRENT stream is about to be cached for every stream (the sub-discriminators, the spectral losses
    and the vocoder each run on their own): finish the fill before anyone can find it.  Once per table and process.  (Found as a
    first-step-only 2e-3 error in one resolution discriminator's gradients when two processes shared a GPU: another stream read
    the n_fft = 2048 twiddles while the kernel that writes them was still queued.)"""
    if t.is_cuda and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(t.device).synchronize()
    return t


def _twiddles(n_fft, device):
    key = (n_fft, str(device))
    t = _TW.get(key)
    if t is None:
        t = torch.empty((n_fft, 2), device=device, dtype=torch.float32)
        call("osp_fft_twiddles", t, n_fft)
        _TW[key] = _publish(t)
    return t


def _padded_window(window, n_fft):
    """torch.stft centres a short window inside n_fft: left pad (n_fft - win_length) // 2."""
    if window is None:
        return None
    wl = window.shape[0]
    if wl == n_fft:
        return window.contiguous()
    # the window is a constant buffer: pad it once per (tensor, version, n_fft), not on every STFT (a fill + a copy per call)
    hit = getattr(window, "_osp_padded", None)
    key = (n_fft, window._version, window.data_ptr())
    if hit is None or hit[0] != key:
        left = (n_fft - wl) // 2
        hit = (key, _publish(F.pad(window.detach(), (left, n_fft - wl - left)).contiguous()))
        try:
            window._osp_padded = hit
        except (AttributeError, RuntimeError):
            pass
    return hit[1]


class _StftMagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, window, n_fft, hop, clamp_min):
        x = x.contiguous()
        B, T = x.shape
        frames = 1 + T // hop
        tw = _twiddles(n_fft, x.device)
        mag = torch.empty((B, frames, n_fft // 2 + 1), device=x.device, dtype=torch.float32)
        call("osp_stft_mag_fwd", x, window, tw, float(clamp_min), mag, B, T, n_fft, hop)
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x, window, tw)
            ctx.cfg = (n_fft, hop, clamp_min)
        return mag

    @staticmethod
    def backward(ctx, dmag):
        x, window, tw = ctx.saved_tensors
        n_fft, hop, clamp_min = ctx.cfg
        B, T = x.shape
        dx = torch.zeros_like(x)
        call("osp_stft_mag_bwd", x, window, tw, float(clamp_min), dmag.contiguous(), dx, B, T, n_fft, hop)
        return dx, None, None, None, None


def stft_magnitude(x, n_fft, hop, win_length=None, window=None, clamp_min=None):
    """|STFT| of x (B,T) with center=True reflect padding -> (B, frames, n_fft/2+1).

    window None = rectangular (DiscriminatorR.spectrogram); clamp_min=1e-7 reproduces disc/loss.py:142
    ``sqrt(clamp(re^2+im^2, 1e-7))``; clamp_min None = plain abs (torchaudio / .abs())."""
    w = _padded_window(window, n_fft)
    return _StftMagFn.apply(x, w, n_fft, hop, -1.0 if clamp_min is None else clamp_min)


class _SpectralLossFn(torch.autograd.Function):
    """(spectral convergence, mean |log y - log x|) of magnitudes x (prediction, carries the gradient) and y (target) from ONE
    reduction launch (osp_spectral_loss_sums) and one backward launch, instead of ~8 + ~12 torch element-wise / reduction ops."""

    @staticmethod
    def forward(ctx, x, y, clip):
        x, y = x.contiguous(), y.contiguous()
        sums = torch.zeros(3, device=x.device, dtype=torch.float32)
        call("osp_spectral_loss_sums", x, y, x.numel(), float(clip), sums)
        ctx.save_for_backward(x, y, sums)
        ctx.clip = float(clip)
        return torch.sqrt(sums[0] / sums[1]), sums[2] / x.numel()

    @staticmethod
    def backward(ctx, g_sc, g_mag):
        x, y, sums = ctx.saved_tensors
        g = torch.stack([g_sc.reshape(()).float(), g_mag.reshape(()).float()])
        dx = torch.empty_like(x)
        call("osp_spectral_loss_bwd", x, y, x.numel(), ctx.clip, sums, g, dx)
        return dx, None, None


_FUSED_SPECTRAL = __import__("os").environ.get("OSP_FUSED_SPECTRAL", "1") != "0"


# ------------------------------------------------------------------------------------------------ MR-STFT loss
class STFTLoss(nn.Module):
    """disc/loss.py:197-228 (+ SpectralConvergenceLoss :231-249, LogSTFTMagnitudeLoss :252-270)."""

    def __init__(self, fft_size=1024, shift_size=120, win_length=600, window="hann_window"):
        super().__init__()
        self.fft_size, self.shift_size, self.win_length = fft_size, shift_size, win_length
        self.register_buffer("window", getattr(torch, window)(win_length))

    def forward(self, x, y):
        xm = stft_magnitude(x, self.fft_size, self.shift_size, self.win_length, self.window, 1e-7)
        ym = stft_magnitude(y, self.fft_size, self.shift_size, self.win_length, self.window, 1e-7)
        if _FUSED_SPECTRAL and xm.is_cuda:
            return _SpectralLossFn.apply(xm, ym.detach(), 0.0)      # (the target side carries no gradient on the path)
        sc = torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
        mag = F.l1_loss(torch.log(ym), torch.log(xm))
        return sc, mag


class MultiResolutionSTFTLoss(nn.Module):
    """disc/loss.py:145-194."""

    def __init__(self, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240),
                 window="hann_window"):
        super().__init__()
        self.stft_losses = nn.ModuleList([STFTLoss(fs, ss, wl, window) for fs, ss, wl in
                                          zip(fft_sizes, hop_sizes, win_lengths)])

    def forward(self, x, y):
        if x.dim() == 3:
            x, y = x.view(-1, x.size(2)), y.view(-1, y.size(2))
        sc_loss, mag_loss = 0.0, 0.0
        for f in self.stft_losses:
            sc, mag = f(x, y)
            sc_loss, mag_loss = sc_loss + sc, mag_loss + mag
        n = len(self.stft_losses)
        return sc_loss / n, mag_loss / n


# ------------------------------------------------------------------------------------------------ mel loss
def mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max):
    """HTK mel filterbank, norm=None, as torchaudio.functional.melscale_fbanks documents it (PARITY UNPINNED:
    torchaudio is not available to pin this against).  -> (n_fft//2+1, n_mels)."""
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


class _MatmulConstFn(torch.autograd.Function):
    """y = x @ W for a constant W (K, N) -- the mel projection -- on the f32 MFMA GEMM."""

    @staticmethod
    def forward(ctx, x, W):
        Kd, N = W.shape
        x2 = x.contiguous().view(-1, Kd)
        y = K.conv_gemm(x2, W, N, cin=Kd, w_strides=(1, 0, N))
        ctx.save_for_backward(W)
        ctx.shape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        (W,) = ctx.saved_tensors
        Kd, N = W.shape
        dx = K.conv_gemm(dy.contiguous().view(-1, N), W, Kd, cin=N, w_strides=(N, 0, 1))
        return dx.view(ctx.shape), None


class _Spectrogram(nn.Module):
    def __init__(self, win_length):
        super().__init__()
        self.register_buffer("window", torch.hann_window(win_length))


class _MelScale(nn.Module):
    def __init__(self, fb):
        super().__init__()
        self.register_buffer("fb", fb)


class _MelSpec(nn.Module):
    """Stand-in for torchaudio.transforms.MelSpectrogram with the same buffer names (spectrogram.window,
    mel_scale.fb) so reference checkpoints load."""

    def __init__(self, sample_rate, n_fft, hop_length, win_length, n_mels, f_min, f_max):
        super().__init__()
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.spectrogram = _Spectrogram(win_length)
        self.mel_scale = _MelScale(mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max))

    def forward(self, x):
        mag = stft_magnitude(x, self.n_fft, self.hop_length, self.win_length, self.spectrogram.window, None)
        return _MatmulConstFn.apply(mag, self.mel_scale.fb)


class MelSpecReconstructionLoss(nn.Module):
    """disc/loss.py:88-120: L1 between log(clip(mel, 1e-7)) of prediction and target."""

    def __init__(self, sample_rate, n_fft, hop_length, win_length, n_mels, f_min, f_max, clip_val=1e-7):
        super().__init__()
        self.clip_val = clip_val
        self.mel_spec = _MelSpec(sample_rate, n_fft, hop_length, win_length, n_mels, f_min, f_max)

    def forward(self, y_hat, y):
        if _FUSED_SPECTRAL and y_hat.is_cuda:
            return _SpectralLossFn.apply(self.mel_spec(y_hat), self.mel_spec(y).detach(), self.clip_val)[1]
        mel_hat = torch.log(torch.clip(self.mel_spec(y_hat), min=self.clip_val))
        mel = torch.log(torch.clip(self.mel_spec(y), min=self.clip_val))
        return F.l1_loss(mel, mel_hat)
