"""Feature-extraction row: oracle vs the reference's outputs (CPU), HIP path vs oracle / goldens (GPU)."""
import numpy as np
import pytest
import torch

CFG = dict(sample_rate=22050, n_feats=100, n_fft=1024, hop_length=256, win_length=1024, f_min=80, f_max=8000)
STAT_KEYS = ("mel_mean", "mel_std", "pitch_mean", "pitch_std", "energy_mean", "energy_std")


def _items(g):
    return [dict(x=np.arange(5 + i) + 1, wav=g[f"wav{i}"], mel=g[f"mel{i}"], energy=g[f"energy{i}"], pitch=g[f"pitch{i}"],
                 sid=None, lid=None) for i in range(4)]


def test_oracle_features_vs_reference_golden(golden):
    from oracle import features as OF
    g = golden("features")
    basis = OF.slaney_mel_basis(22050, 1024, 100, 80, 8000)
    assert np.array_equal(basis, g["basis"])
    # structural properties of the (unpinned) Slaney basis: non-negative triangles, one peak per filter, monotone centres
    assert (basis >= 0).all() and (basis.sum(1) > 0).all()
    assert (np.diff(basis.argmax(1)) > 0).all()
    for i in range(4):
        mel = OF.get_mel(g[f"wav{i}"], basis, 1024, 256, 1024)[0].numpy()
        # 1e-4: torch.stft on different host CPUs (FFT backend / SIMD width) moves the log-mel by ~2e-5
        assert mel.shape == g[f"mel{i}"].shape and np.abs(mel - g[f"mel{i}"]).max() < 1e-4
        e = OF.get_energy(g[f"wav{i}"], mel.shape[-1], 1024, 256, 1024)[0].numpy()
        assert np.allclose(e, g[f"energy{i}"], rtol=1e-5, atol=1e-5)
        et = OF.get_energy(g[f"wav{i}"], mel.shape[-1] - 2, 1024, 256, 1024)[0].numpy()
        assert np.allclose(et, g[f"energy_trim{i}"], rtol=1e-5, atol=1e-5)
    stats = dict(zip(STAT_KEYS, g["stats"].tolist()))
    b = OF.collate(_items(g), 100, stats)
    for k in ("x", "wav", "mel", "x_lengths", "wav_lengths", "mel_lengths", "energies", "pitches"):
        assert np.allclose(b[k], g["collate_" + k], rtol=1e-6, atol=1e-6), k
    # the collate quirk: padded mel positions hold -mean/std, not 0
    assert abs(b["mel"][2, 0, -1] - (-stats["mel_mean"] / stats["mel_std"])) < 1e-6


@pytest.mark.gpu
def test_hip_features_vs_reference_golden(golden):
    from optispeech_amd import features as FE
    g = golden("features")
    assert np.array_equal(FE.slaney_mel_basis(22050, 1024, 100, 80, 8000), g["basis"])
    fe = FE.CommonFeatureExtractor(center=True, **CFG)
    for i in range(4):
        mel = fe.get_mel(g[f"wav{i}"])
        assert mel.shape == g[f"mel{i}"].shape
        # log-mel: absolute tolerance 2e-3 (f32 FFT in LDS vs torch.stft; the log amplifies relative error near the clamp)
        assert np.abs(mel - g[f"mel{i}"]).max() < 2e-3, np.abs(mel - g[f"mel{i}"]).max()
        e = fe.get_energy(g[f"wav{i}"], mel.shape[-1])
        assert np.allclose(e, g[f"energy{i}"], rtol=1e-3, atol=1e-4)
        assert np.allclose(fe.get_energy(g[f"wav{i}"], mel.shape[-1] - 2), g[f"energy_trim{i}"], rtol=1e-3, atol=1e-4)
        assert fe.get_energy(g[f"wav{i}"], mel.shape[-1] + 3).shape[-1] == mel.shape[-1] + 3     # pad branch (reference raises)
    stats = dict(zip(STAT_KEYS, g["stats"].tolist()))
    b = FE.TextWavBatchCollate(100, stats)([{k: (torch.from_numpy(np.asarray(v)) if v is not None else None) for k, v in it.items()}
                                            for it in _items(g)])
    for k in ("x", "wav", "mel", "x_lengths", "wav_lengths", "mel_lengths", "energies", "pitches"):
        assert b[k].is_cuda
        assert np.allclose(b[k].cpu().numpy(), g["collate_" + k], rtol=1e-6, atol=1e-6), k


@pytest.mark.gpu
def test_hip_features_batched_full_size():
    """BASELINE-sized batch (32 x 800 frames) through the device API: batched == per-utterance, energy == ||sqrt(mag^2+eps)||."""
    from optispeech_amd import features as FE
    fe = FE.CommonFeatureExtractor(center=True, **CFG)
    wav = torch.randn(32, 256 * 799 - 768, generator=torch.Generator().manual_seed(0)).clamp(-1, 1).cuda()
    mel, energy = fe.mel_energy_device(wav)
    assert mel.shape == (32, 100, 800) and energy.shape == (32, 800)
    m1, e1 = fe.mel_energy_device(wav[5])
    assert torch.equal(m1[0], mel[5]) and torch.equal(e1[0], energy[5])
    assert torch.isfinite(mel).all() and (mel >= np.log(1e-5) - 1e-6).all()


def test_dataset_reader_and_collate_round_trip(tmp_path, golden):
    """on-disk format (.json + .npz) -> TextWavDataset -> TextWavBatchCollate (CPU device): same batch as the reference's
    collate produced from the same datapoints; unvoiced pitch threshold applied on read"""
    import os
    from types import SimpleNamespace
    from optispeech_amd import features as FE
    g = golden("features")
    stems = []
    for i, it in enumerate(_items(g)):
        stem = os.path.join(tmp_path, f"utt{i}")
        pitch = it["pitch"].copy()
        FE.write_datapoint(stem, it["x"], f"utt {i}", it["wav"], it["mel"], it["energy"], pitch)
        stems.append(stem)
    fl = os.path.join(tmp_path, "train.txt")
    open(fl, "w").write("\n".join(stems) + "\n\n")
    ds = FE.TextWavDataset(1, fl, None, SimpleNamespace(f_min=80), seed=1234)
    assert len(ds) == 4 and ds.uv_threshold == 80 // 3.5 and sorted(ds.file_paths) == sorted(stems)
    ds.file_paths = stems                                        # undo the shuffle for the comparison
    items = [ds[i] for i in range(4)]
    assert items[1]["text"] == "utt 1" and items[2]["x"].dtype == torch.long
    stats = dict(zip(STAT_KEYS, g["stats"].tolist()))
    b = FE.TextWavBatchCollate(100, stats, device="cpu")(items)
    for k in ("x", "wav", "mel", "x_lengths", "wav_lengths", "mel_lengths", "energies", "pitches"):
        assert np.allclose(b[k].numpy(), g["collate_" + k], rtol=1e-6, atol=1e-6), k
    # unvoiced rule
    low = dict(items[0]); stem = os.path.join(tmp_path, "low")
    FE.write_datapoint(stem, [1, 2], "low", g["wav0"], g["mel0"], g["energy0"], np.full_like(g["pitch0"], 10.0))
    assert float(ds.get_datapoint(stem)["pitch"].abs().max()) == 0.0
