"""On-GPU feature extraction + collate: host-side mirror of the reference's preprocessing interface.

  * ``CommonFeatureExtractor.get_mel / get_energy``   optispeech/dataset/feature_extractors/__init__.py:114-147,153-200
  * ``TextWavBatchCollate``                           optispeech/dataset/text_wav_datamodule.py:195-266

Same class / method names, argument meaning and return types (numpy in -> numpy out for get_mel / get_energy, a batch dict
from the collate), but the arithmetic runs in the HIP kernels: osp_stft_mag_fwd (in-LDS FFT, reflect padding on load) and
osp_logmel_energy (epsilon, mel projection, log clamp, frame energy).  Device-resident variants (``*_device``) keep
everything in HBM for an on-line pipeline.  Pitch extraction, loudness normalisation and silence trimming are separate
third-party models in the reference (pyworld / penn / pyloudnorm / Silero-VAD) and stay out of scope.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from ._lib import call
from . import spectral


def _hz_to_mel(f):
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    return f / f_sp if f < min_log_hz else min_log_hz / f_sp + math.log(f / min_log_hz) / (math.log(6.4) / 27.0)


def _mel_to_hz(m):
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp((math.log(6.4) / 27.0) * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sample_rate, n_fft, n_mels, f_min, f_max):
    """The matrix ``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`` returns with its defaults (Slaney mel scale,
    area normalisation, float32) -- feature_extractors/__init__.py:169-172.  (n_mels, 1 + n_fft // 2)."""
    n_freqs = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, float(sample_rate) / 2, n_freqs)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(float(f_min)), _hz_to_mel(float(f_max)), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_freqs), dtype=np.float32)
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels]))[:, None].astype(np.float32)
    return w


def trim_or_pad_to_target_length(data, target_length):
    """utils/model.py:155-165 for 1-D device tensors (the reference's pad branch raises on tensors; we zero-pad)."""
    n = data.shape[-1]
    return data[..., :target_length] if n >= target_length else F.pad(data, (0, target_length - n))


class CommonFeatureExtractor:
    """feature_extractors/__init__.py:18-64,153-200 (the mel / energy part)."""

    def __init__(self, sample_rate, n_feats, n_fft, hop_length, win_length, f_min, f_max, center=True, pitch_extractor=None,
                 device="cuda", **unused):
        assert center, "the reference config uses center=True (configs/data/feature_extractor/default.yaml)"
        self.sample_rate, self.n_feats, self.n_fft, self.hop_length, self.win_length = sample_rate, n_feats, n_fft, hop_length, win_length
        self.f_min, self.f_max, self.center = f_min, f_max, center
        self.pitch_extractor = pitch_extractor
        self.device = torch.device(device)
        self._fbT = None
        self._window = None

    def _consts(self):
        if self._fbT is None:
            fb = slaney_mel_basis(self.sample_rate, self.n_fft, self.n_feats, self.f_min, self.f_max)
            self._fbT = torch.from_numpy(np.ascontiguousarray(fb.T)).to(self.device)
            self._window = torch.hann_window(self.win_length, device=self.device)
        return self._fbT, self._window

    # ---- device-resident API: wav (T,) or (B, T) f32 on the GPU, all rows the same length
    def mel_energy_device(self, wav):
        fbT, window = self._consts()
        y = wav.to(self.device, torch.float32)
        if y.dim() == 1:
            y = y.unsqueeze(0)
        p = int((self.n_fft - self.hop_length) / 2)
        y = F.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1).contiguous()       # :176-181 (data movement)
        mag = spectral.stft_magnitude(y, self.n_fft, self.hop_length, self.win_length, window, None)   # (B, frames, bins)
        B, frames, bins = mag.shape
        mel = torch.empty((B, self.n_feats, frames), device=self.device, dtype=torch.float32)
        energy = torch.empty((B, frames), device=self.device, dtype=torch.float32)
        for b in range(B):
            call("osp_logmel_energy", mag[b], fbT, mel[b], energy[b], frames, bins, self.n_feats, 1e-9, 1e-5)
        return mel, energy

    # ---- the reference's interface (numpy in, numpy out)
    def get_mel(self, wav):
        mel, _ = self.mel_energy_device(torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32)))
        return mel.squeeze().cpu().numpy()

    def get_energy(self, wav, mel_length):
        _, e = self.mel_energy_device(torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32)))
        return trim_or_pad_to_target_length(e.squeeze(0), mel_length).cpu().numpy()


def normalize(data, mu, std):
    """utils/model.py:74-93 (scalar statistics)."""
    return (data - mu) / std


class TextWavBatchCollate:
    """text_wav_datamodule.py:195-266: zero-pad to the batch maxima, THEN normalise (padded positions hold -mean/std).
    ``device`` selects where the batch tensors live (the reference returns CPU tensors and a numpy ``wav``)."""

    def __init__(self, n_feats, data_statistics, do_normalize=True, device="cuda"):
        self.n_feats, self.data_statistics, self.do_normalize = n_feats, data_statistics, do_normalize
        self.device = torch.device(device)

    def __call__(self, batch):
        dev, B = self.device, len(batch)
        t = lambda v, dt: torch.as_tensor(v).to(dev, dt)                              # noqa: E731
        xm = max(it["x"].shape[-1] for it in batch)
        mm = max(it["mel"].shape[-1] for it in batch)
        wm = max(it["wav"].shape[-1] for it in batch)
        x = torch.zeros((B, xm), dtype=torch.long, device=dev)
        wav = torch.zeros((B, wm), dtype=torch.float32, device=dev)
        mel = torch.zeros((B, self.n_feats, mm), dtype=torch.float32, device=dev)
        pitches = torch.zeros((B, mm), dtype=torch.float32, device=dev)
        energies = torch.zeros((B, mm), dtype=torch.float32, device=dev)
        sids, lids = [], []
        for i, it in enumerate(batch):
            x[i, : it["x"].shape[-1]] = t(it["x"], torch.long)
            wav[i, : it["wav"].shape[-1]] = t(it["wav"], torch.float32)
            mel[i, :, : it["mel"].shape[-1]] = t(it["mel"], torch.float32)
            energies[i, : it["energy"].shape[-1]] = t(it["energy"], torch.float32)
            pitches[i, : it["pitch"].shape[-1]] = t(it["pitch"], torch.float32)
            if it.get("sid") is not None:
                sids.append(it["sid"])
            if it.get("lid") is not None:
                lids.append(it["lid"])
        sids = torch.tensor(sids, dtype=torch.long, device=dev) if sids else None
        lids = torch.tensor(lids, dtype=torch.long, device=dev) if lids else None
        if sids is not None:
            assert sids.shape[0] == B, "Not all speaker IDs are provided"
        if lids is not None:
            assert lids.shape[0] == B, "Not all language IDs are provided"
        if self.do_normalize:
            s = self.data_statistics
            wav = wav.clip(-1, 1)
            mel = normalize(mel, s["mel_mean"], s["mel_std"])
            energies = normalize(energies, s["energy_mean"], s["energy_std"])
            pitches = normalize(pitches, s["pitch_mean"], s["pitch_std"])
        ln = lambda k: torch.tensor([it[k].shape[-1] for it in batch], dtype=torch.long, device=dev)   # noqa: E731
        return dict(x=x, wav=wav, mel=mel, x_lengths=ln("x"), wav_lengths=ln("wav"), mel_lengths=ln("mel"), energies=energies,
                    pitches=pitches, sids=sids, lids=lids, x_texts=[it.get("text", "") for it in batch],
                    filepaths=[it.get("filepath", "") for it in batch])


# ------------------------------------------------------------------------------------------------ on-disk dataset
def parse_filelist(filelist_path):
    """dataset/text_wav_datamodule.py:46-49: one datapoint stem per non-empty line."""
    import pathlib
    return [f for f in pathlib.Path(filelist_path).read_text(encoding="utf-8").splitlines() if f.strip()]


def write_datapoint(stem, phoneme_ids, text, wav, mel, energy, pitch, sid=None, lid=None):
    """The on-disk format of tools/preprocess_dataset.py:57-79: ``<stem>.json`` (phoneme_ids, text[, sid, lid]) +
    ``<stem>.npz`` (wav, mel, energy, pitch; no pickles)."""
    import json
    import pathlib
    stem = pathlib.Path(stem)
    meta = {"phoneme_ids": [int(i) for i in phoneme_ids], "text": text}
    if sid is not None:
        meta["sid"] = sid
    if lid is not None:
        meta["lid"] = lid
    with open(stem.with_suffix(".json"), "w", encoding="utf-8") as fh:
        json.dump(meta, fh, ensure_ascii=False)
    np.savez(stem.with_suffix(".npz"), allow_pickle=False, wav=wav, mel=mel, energy=energy, pitch=pitch)


class TextWavDataset(torch.utils.data.Dataset):
    """dataset/text_wav_datamodule.py:133-193 (the reader half): a shuffled file list of ``.json`` + ``.npz`` datapoints;
    pitch values at or below ``f_min // 3.5`` are unvoiced and set to 0 (:163-165)."""

    def __init__(self, num_speakers, filelist_path, text_processor, feature_extractor, seed=None):
        import random
        self.num_speakers, self.text_processor, self.feature_extractor = num_speakers, text_processor, feature_extractor
        self.file_paths = parse_filelist(filelist_path)
        self.uv_threshold = feature_extractor.f_min // 3.5
        random.Random(seed).shuffle(self.file_paths) if seed is not None else random.shuffle(self.file_paths)

    def get_datapoint(self, filepath):
        import json
        import pathlib
        f = pathlib.Path(filepath)
        with open(f.with_suffix(".json"), encoding="utf-8") as fh:
            meta = json.load(fh)
        arrays = np.load(f.with_suffix(".npz"), allow_pickle=False)
        pitch = torch.from_numpy(arrays["pitch"]).clone()
        pitch[pitch <= self.uv_threshold] = 0.0
        return dict(x=torch.LongTensor(meta["phoneme_ids"]), wav=torch.from_numpy(arrays["wav"]), mel=torch.from_numpy(arrays["mel"]),
                    energy=torch.from_numpy(arrays["energy"]), pitch=pitch, sid=meta.get("sid"), lid=meta.get("lid"),
                    text=meta["text"], filepath=str(filepath))

    def __getitem__(self, index):
        return self.get_datapoint(self.file_paths[index])

    def __len__(self):
        return len(self.file_paths)
