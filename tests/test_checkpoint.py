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


def _resume_fixture(pipeline=False):
    from optispeech_amd import rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    cfg = ModelConfig()                                          # dropout / drop-path ON: the RNG position matters

    def prep(m):
        m.optimizers()
        for sch in m.lr_schedulers():                            # no warm-up: every AdamW step is full size (lr 2e-4), so a
            sch.warmup = 0                                       # resume that lost moments / counters moves the weights visibly
            sch.opt.lr = sch.base_lr
        m.pipeline_steps = pipeline
        m.generator.segment_rand01 = torch.rand(2, generator=torch.Generator().manual_seed(1)).cuda()
        return m

    def fresh():
        torch.manual_seed(3)
        torch.cuda.manual_seed(3)
        rng.manual_seed(3, 0)
        return prep(make_optispeech(cfg, batch_size=2, pretraining_steps=0).to("cuda").train())
    return cfg, fresh, prep, synthetic_batch(2, 24, 96, cfg, seed=5, device="cuda")


def _mean_abs_diff(a, b):
    num = den = 0.0
    for k in a:
        if a[k].is_floating_point():
            num += (a[k].double() - b[k].double()).abs().sum().item()
            den += a[k].numel()
    return num / den


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [False, True])
def test_save_checkpoint_resume_is_exact(tmp_path, pipeline):
    """train 4 steps == train 2 -> save_checkpoint -> load_from_checkpoint + load_training_state -> train 2 (weights, AdamW
    moments in the reference layout, schedule, dropout RNG position and stream ids all restored), WITH a negative control: the
    same resume without load_training_state must land measurably elsewhere.  pipeline=True saves while the discriminator phase
    of the last step may still be in flight on its own stream (save_checkpoint joins it first)."""
    from optispeech_amd import precision
    from optispeech_amd.model import OptiSpeech
    precision.set_precision("f32")
    cfg, fresh, prep, batch = _resume_fixture(pipeline)
    a = fresh()
    for i in range(2):
        a.training_step(batch, i)
    path = os.path.join(tmp_path, "resume.ckpt")
    a.save_checkpoint(path)
    gen_state = torch.cuda.get_rng_state()                       # torch's generator (drop-path draws) is the trainer's to save
    for i in range(2, 4):
        a.training_step(batch, i)
    want = {k: v.detach().clone() for k, v in a.state_dict().items()}
    ck = torch.load(path, weights_only=False)
    assert ck["global_step"] == 4 and "generator.vocoder.head.linear_1.weight" in ck["state_dict"]
    mom = ck["osp"]["optimizers"][0]["moments"]["generator.vocoder.head.linear_1.weight"][0]
    assert tuple(mom.shape) == tuple(ck["state_dict"]["generator.vocoder.head.linear_1.weight"].shape)   # reference layout
    assert float(mom.abs().max()) > 0                            # the moments were saved AFTER the updates landed
    got = {}
    for restore in (True, False):
        b = prep(OptiSpeech.load_from_checkpoint(path, config=cfg, strict=True).to("cuda").train())
        b.train_args.pretraining_steps = 0
        if restore:
            b.load_training_state(ck)
        else:
            b.global_step = ck["global_step"]
        torch.cuda.set_rng_state(gen_state)
        for i in range(2, 4):
            b.training_step(batch, i)
        got[restore] = {k: v.detach().clone() for k, v in b.state_dict().items()}
    d_ok, d_bad = _mean_abs_diff(want, got[True]), _mean_abs_diff(want, got[False])
    # two AdamW steps at lr 2e-4 move every weight by ~4e-4: a resume that lost the moments / step counts / dropout position
    # ends ~1e-4 away on average, an exact one differs only where f32 atomics order flips a noise-level gradient sign
    assert d_bad > 2e-5, d_bad
    assert d_ok < 0.1 * d_bad, (d_ok, d_bad)


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


def test_hifigan_generator_state_dict_is_the_reference_schema(golden):
    """The causal HiFi-GAN generator keeps the key names and shapes of the reference modules with
    torch weight_norm applied (`*.conv.weight_g`, `*.deconv.weight_v`, `*.pad_buffer` ...): the golden's key list comes from the
    reference's own modules (tools/make_golden_hifigan.py)."""
    from optispeech_amd.model.hifigan import Generator
    from tests.tools_cfg_hifigan import CFG
    g = golden("hifigan_small")
    sd = Generator(**CFG).state_dict()
    assert sorted(sd) == sorted(g["keys"].tolist())                       # (order inside a module differs: buffers last here)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(g["w_" + k].shape), k
