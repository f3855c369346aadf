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


def _resume_fixture(pipeline=False, backbone="convnext"):
    from optispeech_amd import rng
    from optispeech_amd.config import ModelConfig, make_optispeech, synthetic_batch
    cfg = ModelConfig(backbone=backbone)                         # dropout / drop-path ON: the RNG position matters

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
@pytest.mark.parametrize("pipeline,backbone", [(False, "convnext"), (True, "convnext"), (False, "transformer"), (False, "lightspeech"),
                                               (False, "leanspeech"), (False, "conformer")])
def test_save_checkpoint_resume_is_exact(tmp_path, pipeline, backbone):
    """train 4 steps == train 2 -> save_checkpoint -> load_from_checkpoint + load_training_state -> train 2 (weights, AdamW
    moments in the reference layout, schedule, dropout RNG position and stream ids all restored), WITH a negative control: the
    same resume without load_training_state must land measurably elsewhere.  pipeline=True saves while the discriminator phase
    of the last step may still be in flight on its own stream (save_checkpoint joins it first)."""
    from optispeech_amd import precision
    from optispeech_amd.model import OptiSpeech
    precision.set_precision("f32")
    cfg, fresh, prep, batch = _resume_fixture(pipeline, backbone)
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


@pytest.mark.gpu
def test_lightning_checkpoint_with_optimizer_state_resumes_like_torch_adamw(tmp_path, golden):
    """A checkpoint in the layout Lightning writes for the reference module (state_dict in the reference key order,
    ``optimizer_states`` = two torch AdamW state dicts indexed in ``module.parameters()`` order -- the order fixture was taken from
    the reference modules --, ``lr_schedulers``, pickled hyper-parameters that reference classes this process cannot import):
    ``load_from_checkpoint`` + ``load_lightning_training_state`` restore weights, both moment sets, step counts and learning rates,
    and the NEXT update equals torch.optim.AdamW's next update on the reference-layout tensors."""
    import functools
    import sys
    import types
    from collections import OrderedDict
    from optispeech_amd.config import make_optispeech
    from optispeech_amd.model.optispeech import OptiSpeech
    from tests.test_gpu_generator import _small_cfg
    order = golden("ref_param_order")
    dev = "cuda"
    torch.manual_seed(3)
    src = make_optispeech(_small_cfg(), batch_size=2, pretraining_steps=0).to(dev)
    sd = src.state_dict()
    gen = torch.Generator(device=dev).manual_seed(11)
    ref_params, opts = {}, []
    for prefix, names in (("generator.", order["generator_params"]), ("discriminator.", order["discriminator_params"])):
        ps = [torch.nn.Parameter(sd[prefix + str(k)].detach().clone().float()) for k in names]
        ref_params[prefix] = ps
        opt = torch.optim.AdamW(ps, lr=1.5e-4, betas=(0.8, 0.99), weight_decay=0.01, eps=1e-8)
        for _ in range(2):                                                    # two steps: non-trivial moments, step = 2
            for p in ps:
                p.grad = torch.randn(p.shape, device=dev, generator=gen) * 0.01
            opt.step()
        opts.append(opt)
    state = OrderedDict()
    for prefix, keys, names in (("generator.", order["generator_state_keys"], order["generator_params"]),
                                ("discriminator.", order["discriminator_state_keys"], order["discriminator_params"])):
        upd = {str(k): p.detach().cpu() for k, p in zip(names, ref_params[prefix])}
        for k in keys:
            k = str(k)
            if prefix + k in sd:                                              # MR-STFT windows etc. are not in our schema
                state[prefix + k] = upd.get(k, sd[prefix + k].detach().cpu())
    # hyper-parameters the way Hydra leaves them: partials of reference classes
    fake = types.ModuleType("optispeech_ref_only_module")
    fake.OptiSpeechGenerator = type("OptiSpeechGenerator", (), {"__module__": "optispeech_ref_only_module"})
    sys.modules["optispeech_ref_only_module"] = fake
    ckpt = {"epoch": 7, "global_step": 4, "pytorch-lightning_version": "2.2.1", "state_dict": state, "loops": {}, "callbacks": {},
            "optimizer_states": [o.state_dict() for o in opts],
            "lr_schedulers": [{"last_epoch": 2, "_step_count": 3, "base_lrs": [2e-4]}, {"last_epoch": 2, "_step_count": 3, "base_lrs": [2e-4]}],
            "hyper_parameters": {"generator": functools.partial(fake.OptiSpeechGenerator)}}
    path = tmp_path / "epoch=7.ckpt"
    try:
        torch.save(ckpt, path)
    finally:
        del sys.modules["optispeech_ref_only_module"]
    m = OptiSpeech.load_from_checkpoint(str(path), config=_small_cfg(), strict=True).to(dev).train()
    assert m.ckpt_loaded_epoch == 7
    m.load_lightning_training_state(str(path))
    og, od = m.optimizers()
    assert (og.step_count, od.step_count, m.global_step) == (2, 2, 4) and abs(og.lr - 1.5e-4) < 1e-12
    assert [s.last_step for s in m.lr_schedulers()] == [2, 2]
    sd2 = m.state_dict()
    for k, v in state.items():
        assert torch.equal(sd2[k].cpu(), v), k
    # moments, through the module's own layout mapping
    for opt, topt, prefix, names in ((og, opts[0], "generator.", order["generator_params"]), (od, opts[1], "discriminator.", order["discriminator_params"])):
        tstate = {prefix + str(k): topt.state[p] for k, p in zip(names, ref_params[prefix])}
        for key, to_ref, _, a, b in m._moment_views(opt):
            ea = to_ref(a) if to_ref else a
            assert torch.allclose(ea, tstate[key]["exp_avg"], rtol=0, atol=0), key
            eb = to_ref(b) if to_ref else b
            assert torch.equal(eb, tstate[key]["exp_avg_sq"]), key
    # the next update: same gradients on both sides (reference layout -> native through the mapping), no clipping
    for opt, topt, prefix, names in ((og, opts[0], "generator.", order["generator_params"]), (od, opts[1], "discriminator.", order["discriminator_params"])):
        gref = {prefix + str(k): torch.randn(p.shape, device=dev, generator=gen) * 0.01 for k, p in zip(names, ref_params[prefix])}
        for k, p in zip(names, ref_params[prefix]):
            p.grad = gref[prefix + str(k)]
        topt.step()
        opt.zero_grad()
        for (key, _, to_native, _, _), p in zip(m._moment_views(opt), opt.arena.params):
            g = gref[key]
            p.grad.copy_(to_native(g) if to_native else g)
        opt.step(max_norm=None)
    torch.cuda.synchronize()
    sd3 = m.state_dict()
    for prefix, names in (("generator.", order["generator_params"]), ("discriminator.", order["discriminator_params"])):
        for k, p in zip(names, ref_params[prefix]):
            torch.testing.assert_close(sd3[prefix + str(k)], p.detach(), rtol=2e-6, atol=1e-8, msg=lambda m_: f"{prefix}{k}: {m_}")
