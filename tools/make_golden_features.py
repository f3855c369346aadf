#!/usr/bin/env python3
"""Golden fixtures for the feature-extraction row by RUNNING THE REFERENCE's own functions here.

    python tools/make_golden_features.py        # writes tests/golden/features.npz

Stubs: librosa is not installable in the build container -> `librosa.filters.mel` is served by
oracle.features.slaney_mel_basis (the matrix is stored in the fixture, so the reference code and the build multiply by
the same basis; the basis formula itself stays "parity unpinned"), `librosa.util.normalize` by peak normalisation;
torchaudio / pyloudnorm / onnxruntime-backed silence detector are import-only stand-ins (not on the tested path).
Only inputs and the reference's numerical outputs are stored; no reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from oracle import features as OF  # noqa: E402
from tools.make_golden import install_stubs, _stub, _Any  # noqa: E402

install_stubs()
_stub("librosa", util=types.SimpleNamespace(normalize=lambda w: w / np.abs(w).max()), load=_Any(), effects=_Any())
_stub("librosa.filters", mel=lambda sr, n_fft, n_mels, fmin, fmax: OF.slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax))
sys.modules["librosa"].filters = sys.modules["librosa.filters"]
_stub("pyloudnorm", Meter=_Any(), normalize=_Any())
_stub("pyworld")
_stub("optispeech.text", TextProcessor=_Any())
_stub("torchaudio", transforms=_Any(), functional=types.SimpleNamespace(highpass_biquad=_Any(), lowpass_biquad=_Any()))
_stub("torchaudio.functional", highpass_biquad=_Any(), lowpass_biquad=_Any())
_stub("optispeech.dataset.feature_extractors.norm_audio", make_silence_detector=_Any(), trim_audio=_Any())

from optispeech.dataset.feature_extractors import CommonFeatureExtractor  # noqa: E402
from optispeech.dataset.text_wav_datamodule import TextWavBatchCollate  # noqa: E402

CFG = dict(sample_rate=22050, n_feats=100, n_fft=1024, hop_length=256, win_length=1024, f_min=80, f_max=8000, center=True)
fe = CommonFeatureExtractor(pitch_extractor=None, **CFG)
rng = np.random.default_rng(7)
out = {"basis": OF.slaney_mel_basis(22050, 1024, 100, 80, 8000)}
lengths = [22050, 30000, 4096 + 123, 256 * 40]
items = []
for i, T in enumerate(lengths):
    t = np.arange(T) / 22050.0
    wav = (0.5 * np.sin(2 * np.pi * (110.0 * (i + 1)) * t) * np.exp(-t) + 0.05 * rng.standard_normal(T)).astype(np.float32)
    wav = wav / np.abs(wav).max()
    mel = fe.get_mel(wav)
    energy = fe.get_energy(wav, mel.shape[-1])
    energy_trim = fe.get_energy(wav, mel.shape[-1] - 2)      # (the pad branch of the reference raises: np.concatenate on a tensor)
    out[f"wav{i}"], out[f"mel{i}"], out[f"energy{i}"], out[f"energy_trim{i}"] = wav, mel, energy, energy_trim
    items.append(dict(x=torch.arange(5 + i) + 1, wav=torch.from_numpy(wav), mel=torch.from_numpy(mel),
                      energy=torch.from_numpy(energy), pitch=torch.from_numpy((100 + 10 * rng.random(mel.shape[-1])).astype(np.float32)),
                      sid=None, lid=None, text="", filepath=""))
    out[f"pitch{i}"] = items[-1]["pitch"].numpy()
stats = dict(mel_mean=-5.536622, mel_std=2.116101, pitch_mean=206.0, pitch_std=53.6, energy_mean=21.1, energy_std=18.0)
b = TextWavBatchCollate(n_feats=100, data_statistics=stats)(items)
for k in ("x", "wav", "mel", "x_lengths", "wav_lengths", "mel_lengths", "energies", "pitches"):
    out["collate_" + k] = np.asarray(b[k])
out["stats"] = np.array([stats[k] for k in ("mel_mean", "mel_std", "pitch_mean", "pitch_std", "energy_mean", "energy_std")])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "features.npz"), **out)
print("wrote features.npz", {k: v.shape for k, v in out.items() if k.startswith("mel")})
