"""CPU restatement of the reference's wav -> (mel, energy) feature extraction and collate normalisation.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and tools/ that build fixtures); the product (optispeech_amd/) never
imports this package.

Follows
  * CommonFeatureExtractor.get_mel           optispeech/dataset/feature_extractors/__init__.py:158-200
  * FeatureExtractor.get_energy              optispeech/dataset/feature_extractors/__init__.py:114-147
  * spectral_normalize_torch (log clamp 1e-5) optispeech/utils/audio.py:23-34
  * trim_or_pad_to_target_length             optispeech/utils/model.py:155-165
  * TextWavBatchCollate (+ normalize)        optispeech/dataset/text_wav_datamodule.py:195-266, utils/model.py:74-93
Pinned by tests/golden/features.npz, produced by running those reference functions in the build container
(tools/make_golden_features.py).  One dependency is absent there: librosa (pinned 0.10.x in the reference's
requirements).  `slaney_mel_basis` restates librosa.filters.mel(htk=False, norm="slaney") from its published definition;
the fixture generator hands the same matrix to the reference code, so the STFT / epsilon / log / energy / padding /
normalisation semantics are pinned by the reference run, while the mel BASIS itself is "parity unpinned"
(no librosa binary to compare with).
"""
import numpy as np
import torch


def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sample_rate, n_fft, n_mels, f_min, f_max):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults htk=False, norm='slaney', dtype=float32
    -> (n_mels, 1 + n_fft//2).  PARITY UNPINNED (librosa is not installable here)."""
    n_freqs = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, float(sample_rate) / 2, n_freqs)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_freqs), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis].astype(np.float32)
    return weights


def magnitudes(wav, n_fft, hop, win, center=True):
    """feature_extractors/__init__.py:120-143 / :176-196: reflect pad (n_fft-hop)/2 on both sides, torch.stft (hann,
    `center` reflect padding on top of it), sqrt(re^2 + im^2 + 1e-9).  wav (T,) or (B,T) -> (B, bins, frames)."""
    y = torch.as_tensor(wav, dtype=torch.float32)
    if y.dim() == 1:
        y = y.unsqueeze(0)
    p = int((n_fft - hop) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win),
                                         center=center, pad_mode="reflect", normalized=False, onesided=True,
                                         return_complex=True))
    return torch.sqrt(spec.pow(2).sum(-1) + 1e-9)


def get_mel(wav, basis, n_fft, hop, win, center=True):
    """log(clamp(basis @ magnitudes, 1e-5)) -> (B, n_mels, frames)   (:197-199, utils/audio.py:23-34)"""
    mag = magnitudes(wav, n_fft, hop, win, center)
    return torch.log(torch.clamp(torch.matmul(torch.as_tensor(basis), mag), min=1e-5))


def get_energy(wav, mel_length, n_fft, hop, win, center=True):
    """L2 norm of the magnitudes over frequency, trimmed / zero-padded to mel_length (:144-146) -> (B, mel_length)"""
    e = torch.norm(magnitudes(wav, n_fft, hop, win, center), dim=1)
    if e.shape[-1] >= mel_length:
        return e[..., :mel_length]
    return torch.nn.functional.pad(e, (0, mel_length - e.shape[-1]))


def collate(items, n_feats, stats, do_normalize=True):
    """TextWavBatchCollate.__call__ (text_wav_datamodule.py:201-266): zero-pad to the batch maxima, THEN normalise --
    so padded positions hold (0 - mean) / std, not 0.  items: dicts with x, wav, mel (n_feats, T), energy, pitch."""
    B = len(items)
    xm, mm, wm = (max(np.shape(it[k])[-1] for it in items) for k in ("x", "mel", "wav"))
    x = np.zeros((B, xm), np.int64)
    wav = np.zeros((B, wm), np.float32)
    mel = np.zeros((B, n_feats, mm), np.float32)
    pit, ene = np.zeros((B, mm), np.float32), np.zeros((B, mm), np.float32)
    for i, it in enumerate(items):
        x[i, : len(it["x"])] = it["x"]
        wav[i, : len(it["wav"])] = it["wav"]
        mel[i, :, : it["mel"].shape[-1]] = it["mel"]
        ene[i, : len(it["energy"])] = it["energy"]
        pit[i, : len(it["pitch"])] = it["pitch"]
    if do_normalize:
        wav = wav.clip(-1, 1)
        mel = (mel - stats["mel_mean"]) / stats["mel_std"]
        ene = (ene - stats["energy_mean"]) / stats["energy_std"]
        pit = (pit - stats["pitch_mean"]) / stats["pitch_std"]
    return {"x": x, "wav": wav, "mel": mel.astype(np.float32), "energies": ene.astype(np.float32),
            "pitches": pit.astype(np.float32),
            "x_lengths": np.array([len(it["x"]) for it in items], np.int64),
            "wav_lengths": np.array([len(it["wav"]) for it in items], np.int64),
            "mel_lengths": np.array([it["mel"].shape[-1] for it in items], np.int64)}
