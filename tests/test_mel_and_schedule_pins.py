"""Pins for restatements that no reference artefact could pin in round 1 (torchaudio / librosa are absent), against the
independent published implementations that ARE in the image (`transformers`):

  * HTK mel filterbank of torchaudio's MelSpectrogram (reference call site vocoder/wavenext/disc/loss.py:94-107) and the
    Slaney-normalised librosa basis of the feature extractor (dataset/feature_extractors/__init__.py:114-147)
    vs transformers.audio_utils.mel_filter_bank (its docstring states it reproduces torchaudio / librosa);
  * the cosine-with-warm-up schedule (configs/model/scheduler/cosine_with_warmup.yaml -> transformers
    get_cosine_schedule_with_warmup, base_lightning_module.py:56-66) vs transformers itself on a torch optimiser.
"""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def tfb():
    from transformers.audio_utils import mel_filter_bank
    return mel_filter_bank


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(22050, 1024, 100, 80, 8000), (24000, 2048, 80, 0, 12000)])
def test_htk_mel_filterbank_matches_transformers(tfb, sr, n_fft, n_mels, fmin, fmax):
    from oracle import losses as OL
    from optispeech_amd import spectral
    want = tfb(n_fft // 2 + 1, n_mels, fmin, fmax, sr, norm=None, mel_scale="htk")
    for got in (OL.mel_filterbank(sr, n_fft, n_mels, fmin, fmax), spectral.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)):
        got = got.numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 5e-5, np.abs(got - want).max()          # f32 vs f64 evaluation of the same triangles
        assert np.array_equal(got > 1e-4, want > 1e-4) or np.abs(got - want)[(got > 1e-4) != (want > 1e-4)].max() < 5e-5


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(22050, 1024, 100, 80, 8000), (24000, 2048, 80, 0, 12000)])
def test_slaney_mel_basis_matches_transformers(tfb, sr, n_fft, n_mels, fmin, fmax):
    from oracle import features as OF
    from optispeech_amd import features as PF
    want = tfb(n_fft // 2 + 1, n_mels, fmin, fmax, sr, norm="slaney", mel_scale="slaney").T      # librosa layout (n_mels, bins)
    for got in (OF.slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax), PF.slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)):
        got = np.asarray(got, dtype=np.float64)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-7, np.abs(got - want).max()


def test_cosine_warmup_schedule_matches_transformers():
    from transformers import get_cosine_schedule_with_warmup
    from optispeech_amd.optim import cosine_warmup_factor
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=2e-4)
    warm, total = 1000, 50_000
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
    for step in range(3001):
        want = sch.get_last_lr()[0]
        got = 2e-4 * cosine_warmup_factor(step, warm, total)
        assert abs(got - want) <= 1e-12 + 1e-9 * want, (step, got, want)
        opt.step()
        sch.step()
    for step in (0, 1, 500, 1000, 1001, 25_000, 49_999, 50_000, 60_000):
        lam = sch.lr_lambdas[0]
        assert abs(cosine_warmup_factor(step, warm, total) - lam(step)) < 1e-12, step


def test_schedule_object_steps_like_lambdalr():
    """CosineWarmupSchedule (what configure_optimizers returns with interval 'step') against torch's LambdaLR driving the
    same lambda: lr before the first step, after k steps, and the (opt.lr, last_step) pair a checkpoint stores."""
    from optispeech_amd.optim import CosineWarmupSchedule, cosine_warmup_factor

    class Opt:
        lr = 2e-4
    o = Opt()
    s = CosineWarmupSchedule(o, num_warmup_steps=10, num_training_steps=100)
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.SGD([p], lr=2e-4)
    ref = torch.optim.lr_scheduler.LambdaLR(ref_opt, lambda k: cosine_warmup_factor(k, 10, 100))
    for k in range(120):
        assert abs(o.lr - ref.get_last_lr()[0]) < 1e-15, k
        assert s.get_last_lr() == [o.lr] and s.last_step == k
        ref_opt.step()
        ref.step()
        s.step()
