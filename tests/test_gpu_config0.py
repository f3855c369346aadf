"""BASELINE.json configs[0] -- "ConvNeXt backbone (optispeech.yaml), LJSpeech single-speaker, batch=8, 1 step (plumbing)":
the whole input side of the path in front of the training step, as the reference wires it
(dataset/text_wav_datamodule.py:133-266 -> base_lightning_module.py:24-45,78-126):

    wave -> feature extraction (mel, energy; on-GPU STFT) -> on-disk datapoints (.json + .npz) -> TextWavDataset (shuffled file
    list, unvoiced-pitch rule) -> TextWavBatchCollate (zero-pad, clip, z-normalise; CPU tensors and a HOST NUMPY ``wav`` exactly as
    the reference's collate returns them) -> OptiSpeech.training_step(batch, 0) at the full BASELINE model size, B = 8.

Checked against the CPU oracle on the same weights and the same collated batch: MAS durations and segment starts EXACT, the
ground-truth wave segment exact, the four acoustic losses to 1e-4, wav_hat to 1e-3; then the full GAN step must produce finite
logs and move both parameter arenas.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
STATS = dict(mel_mean=-5.5366, mel_std=2.1161, pitch_mean=200.0, pitch_std=50.0, energy_mean=30.0, energy_std=20.0)   # ljspeech.yaml:23-24 (mel)


def test_config0_reader_collate_training_step_b8(tmp_path):
    from oracle import generator as OG
    from oracle import schema as S
    from optispeech_amd import features as FE, precision
    from optispeech_amd.config import FeatureExtractorArgs, ModelConfig, make_optispeech
    precision.set_precision("f32")
    B = 8
    fa = FeatureExtractorArgs()
    fe = FE.CommonFeatureExtractor(sample_rate=fa.sample_rate, n_feats=fa.n_feats, n_fft=fa.n_fft, hop_length=fa.hop_length,
                                   win_length=fa.win_length, f_min=fa.f_min, f_max=fa.f_max, center=True)
    g = torch.Generator().manual_seed(8)
    stems = []
    for i in range(B):
        n_frames = int(torch.randint(80, 131, (1,), generator=g))
        n_tok = int(torch.randint(12, 31, (1,), generator=g))
        t = torch.arange(n_frames * fa.hop_length - 7 * i) / fa.sample_rate                     # (wav length not a hop multiple)
        wav = (0.4 * torch.sin(2 * np.pi * (110.0 + 20 * i) * t) + 0.05 * torch.randn(t.shape[0], generator=g)).clamp(-1, 1).numpy()
        mel = fe.get_mel(wav)
        energy = fe.get_energy(wav, mel.shape[-1])
        pitch = np.where(np.arange(mel.shape[-1]) % 7 == 0, 10.0, 110.0 + 20 * i).astype(np.float32)   # some unvoiced frames
        stem = os.path.join(tmp_path, f"LJ{i:03d}")
        FE.write_datapoint(stem, torch.randint(1, 159, (n_tok,), generator=g).tolist(), f"sentence {i}", wav, mel, energy, pitch)
        stems.append(stem)
    fl = os.path.join(tmp_path, "train.txt")
    open(fl, "w").write("\n".join(stems) + "\n")
    ds = FE.TextWavDataset(1, fl, None, SimpleNamespace(f_min=fa.f_min), seed=1234)
    assert len(ds) == B
    items = [ds[i] for i in range(B)]
    batch = FE.TextWavBatchCollate(fa.n_feats, STATS, device="cpu")(items)
    batch["wav"] = batch["wav"].numpy()                         # the reference's collate hands `wav` over as host numpy (:253-266)
    assert batch["x"].shape[0] == B and batch["sids"] is None and isinstance(batch["wav"], np.ndarray)
    assert float(batch["pitches"].min()) == pytest.approx((0.0 - STATS["pitch_mean"]) / STATS["pitch_std"])   # unvoiced -> 0 -> normalised

    cfg = ModelConfig().no_dropout()                             # BASELINE widths; deterministic for the oracle comparison
    torch.manual_seed(21)
    m = make_optispeech(cfg, batch_size=B, pretraining_steps=0).to("cuda").train()
    W = S.make_weights(S.generator_schema(S.Cfg()), 99)
    m.generator.load_state_dict({k[len("generator."):]: v for k, v in W.items()})
    rand01 = torch.rand(B, generator=g)
    m.generator.segment_rand01 = rand01.to("cuda")
    want = OG.generator_forward({k: v.clone() for k, v in W.items()}, batch, rand01=rand01, keep=True)
    out = m._process_batch(batch)
    assert np.array_equal(out["_aux"]["durations"].cpu().numpy(), want["durations"].numpy()), "MAS durations differ"
    assert np.array_equal(out["start_idx"].cpu().numpy(), want["start_idx"].numpy())
    # the ground-truth segment: rows [start, start + 64) of the hop-framed (zero-padded) host wave
    wav = torch.from_numpy(batch["wav"])
    hop, seg = fa.hop_length, cfg.segment_size
    wav = torch.nn.functional.pad(wav, (0, (-wav.shape[1]) % hop))
    ref_seg = torch.stack([wav[b, int(s) * hop:(int(s) + seg) * hop] for b, s in enumerate(want["start_idx"])])
    assert torch.equal(out["wav"].cpu(), ref_seg)
    for k in ("loss", "align_loss", "duration_loss", "pitch_loss", "energy_loss"):
        a, b_ = float(out[k].detach()), float(want[k].detach())
        assert abs(a - b_) <= 1e-4 * max(1.0, abs(b_)), (k, a, b_)
    werr = ((out["wav_hat"].detach().cpu() - want["wav_hat"].detach()).abs().max() / want["wav_hat"].detach().abs().max()).item()
    assert werr < 1e-3, werr
    # one full GAN step through the public entry point
    og, od = m.optimizers()
    for sch in m.lr_schedulers():
        sch.warmup = 0
        sch.opt.lr = sch.base_lr
    w0 = [o.arena.data.clone() for o in (og, od)]
    assert m.training_step(batch, 0) is None                     # manual optimisation: returns None (base_lightning_module.py:78)
    logs = m.fetch_logs()
    assert len(logs) >= 14 and all(np.isfinite(v) for v in logs.values()), logs
    assert abs(logs["total_loss/train_am_loss"] - float(want["loss"].detach())) <= 1e-4 * abs(float(want["loss"].detach()))
    torch.cuda.synchronize()
    assert not torch.equal(og.arena.data, w0[0]) and not torch.equal(od.arena.data, w0[1])
    assert (og.step_count, od.step_count, m.global_step) == (1, 1, 2)
