"""reference-schema checkpoints (export / import / resume) and the ONNX-graph I/O signature."""
import os

import numpy as np
import pytest
import torch


def _small():
    from optispeech_amd.config import ModelConfig, make_optispeech
    return make_optispeech(ModelConfig(), batch_size=2, pretraining_steps=0)


def test_state_dict_is_the_reference_schema():
    """keys and shapes of state_dict() == the reference's (oracle.schema lists them from the reference modules)"""
    from oracle import schema as S
    m = _small()
    sd = m.state_dict()
    want = dict(S.generator_schema(S.Cfg()))
    want.update(S.discriminator_schema())
    for k, shape in want.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(shape), (k, tuple(sd[k].shape), tuple(shape))
    extra = [k for k in sd if k not in want and "melspec_loss" not in k and "window" not in k and "mr_stft" not in k]
    assert not extra, extra[:5]


def test_state_dict_round_trip_cpu(tmp_path):
    from optispeech_amd.model import OptiSpeech
    a = _small()
    with torch.no_grad():
        for p in a.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    path = os.path.join(tmp_path, "x.ckpt")
    torch.save({"state_dict": a.state_dict(), "epoch": 7}, path)
    b = OptiSpeech.load_from_checkpoint(path, strict=True)
    assert b.ckpt_loaded_epoch == 7
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


@pytest.mark.gpu
def test_save_checkpoint_resume_is_exact(tmp_path):
    """train 2 steps == train 1 step -> save_checkpoint -> load_from_checkpoint + load_training_state -> train 1 step
    (weights, AdamW moments in the reference layout, schedule, dropout RNG position all restored)."""
    from optispeech_amd import precision, rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    from optispeech_amd.model import OptiSpeech
    precision.set_precision("f32")
    cfg = ModelConfig()

    def fresh():
        torch.manual_seed(3)
        rng.manual_seed(3, 0)
        m = make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train()
        m.optimizers()
        return m
    batch = synthetic_batch(2, 24, 96, cfg, seed=5, device="cuda")
    r01 = torch.rand(2, generator=torch.Generator().manual_seed(1))
    a = fresh()
    a.generator.segment_rand01 = r01
    a.training_step(batch, 0)
    path = os.path.join(tmp_path, "resume.ckpt")
    a.save_checkpoint(path)
    a.training_step(batch, 1)
    ck = torch.load(path, weights_only=False)
    assert ck["global_step"] == 2 and "generator.vocoder.head.linear_1.weight" in ck["state_dict"]
    mom = ck["osp"]["optimizers"][0]["moments"]["generator.vocoder.head.linear_1.weight"][0]
    assert tuple(mom.shape) == tuple(ck["state_dict"]["generator.vocoder.head.linear_1.weight"].shape)   # reference layout
    b = OptiSpeech.load_from_checkpoint(path, config=cfg, strict=True).to("cuda").train()
    b.train_args.pretraining_steps = 0
    b.load_training_state(ck)
    b.generator.segment_rand01 = r01
    b.training_step(batch, 1)
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.allclose(va, vb, rtol=1e-5, atol=1e-6), (k, (va - vb).abs().max().item())


@pytest.mark.gpu
def test_onnx_io_signature_matches_synthesise():
    from optispeech_amd import precision
    from optispeech_amd.config import ModelConfig, make_optispeech
    from optispeech_amd.values import InferenceInputs
    precision.set_precision("f32")
    torch.manual_seed(0)
    m = make_optispeech(ModelConfig()).to("cuda").eval()
    x = torch.randint(1, 150, (3, 20))
    xl = torch.tensor([20, 11, 17])
    x = x * (torch.arange(20)[None] < xl[:, None])
    wav, wav_lengths, durations = m.onnx_io(x.numpy(), xl.numpy(), np.array([1.1, 1.6, 1.2], np.float32))
    out = m.synthesise(InferenceInputs(clean_text="", x=x, x_lengths=xl, d_factor=1.1, p_factor=1.6, e_factor=1.2))
    assert torch.equal(torch.as_tensor(out.wav).cpu(), wav.cpu()) and torch.equal(torch.as_tensor(out.durations).cpu(), durations.cpu())
    assert torch.equal(torch.as_tensor(out.wav_lengths).cpu(), wav_lengths.cpu())
