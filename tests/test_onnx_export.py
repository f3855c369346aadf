"""SURVEY.md 8f row 3, the ``.onnx`` writer (reference: optispeech/onnx/export.py:20-125, consumer onnx/infer.py:24-145).
The written FILE is read back with an independent protobuf reader and evaluated with a numpy interpreter of the ONNX operator
semantics (tests/_onnx_numpy.py: neither onnx nor onnxruntime is in the image) and must reproduce ``synthesise``:
the CPU oracle here, the HIP path in the -m gpu test, and the reference-run golden ``synth_small``."""
import json

import numpy as np
import pytest
import torch

from tests import _onnx_numpy as ON


def _small_model(golden_seed=None, dur_bias=None):
    from oracle import schema as S
    from optispeech_amd.config import make_optispeech
    from tests.test_gpu_generator import _small_cfg
    m = make_optispeech(_small_cfg(), batch_size=2).eval()
    W = S.make_weights(S.generator_schema(S.SMALL), 31 if golden_seed is None else golden_seed)
    W["generator.duration_predictor.linear.bias"].fill_(1.2 if dur_bias is None else dur_bias)
    m.generator.load_state_dict({k[len("generator."):]: v for k, v in W.items()})
    return m, W


def test_onnx_file_structure_and_metadata(tmp_path):
    from optispeech_amd.onnx_export import export_as_onnx
    m, W = _small_model()
    path = export_as_onnx(m, str(tmp_path / "model.onnx"))
    model = ON.parse_model(open(path, "rb").read())
    assert model["ir_version"] == 8 and model["opset"] == {"": 16}                       # export.py:16 DEFAULT_OPSET
    g = model["graph"]
    assert [(n, e) for n, e, _ in g["inputs"]] == [("x", 7), ("x_lengths", 7), ("scales", 1)]      # export.py:40-44 (int64, int64, float)
    assert [n for n, _, _ in g["outputs"]] == ["wav", "wav_lengths", "durations"]       # export.py:46
    assert [d for _, _, d in g["inputs"]] == [["batch_size", "time"], ["batch_size"], [3]]
    # every weight of the generator the inference graph needs is an initializer under its reference state-dict key
    sd = {"generator." + k for k in m.generator.state_dict() if "alignment_module" not in k}      # export.py:71 deletes the aligner
    have = set(g["inits"])
    conv_or_norm = {k for k in sd if ("linear_" not in k and "pwconv" not in k and not k.endswith("linear.weight"))}
    assert conv_or_norm <= have, sorted(conv_or_norm - have)[:5]
    info = json.loads(model["metadata"]["inference"])                                    # infer.py:39-51 reads these keys
    for k in ("name", "sample_rate", "inference_args", "text_processor", "speakers", "languages"):
        assert k in info
    assert info["sample_rate"] == 22050 and set(info["inference_args"]) == {"d_factor", "p_factor", "e_factor"}
    # every operator is a standard-domain op of opset <= 16
    ops = {n["op"] for n in g["nodes"]}
    assert ops <= {"Identity", "Add", "Sub", "Mul", "Div", "Max", "Less", "Equal", "Where", "Cast", "Shape", "Gather", "Range",
                   "Unsqueeze", "Squeeze", "Sin", "Cos", "Exp", "Sqrt", "Ceil", "Erf", "Relu", "Concat", "Transpose", "Conv",
                   "ReduceMean", "ReduceMax", "ReduceSum", "CumSum", "MatMul", "Softmax", "Reshape", "Clip"}, ops


@pytest.mark.parametrize("lens,scales", [([24, 17], (1.0, 1.0, 1.0)), ([9], (1.3, 1.6, 0.7)), ([30, 1, 12], (0.8, 1.0, 1.2))])
def test_onnx_graph_reproduces_oracle_synthesise(tmp_path, lens, scales):
    from oracle import generator as OG
    from optispeech_amd.onnx_export import export_as_onnx
    m, W = _small_model()
    path = export_as_onnx(m, str(tmp_path / "model.onnx"))
    model = ON.parse_model(open(path, "rb").read())
    xl = torch.tensor(lens)
    Tt = int(xl.max())
    x = torch.randint(1, 159, (len(lens), Tt), generator=torch.Generator().manual_seed(3)) * (torch.arange(Tt)[None] < xl[:, None])
    want = OG.synthesise({k: v.clone() for k, v in W.items()}, x, xl, d_factor=scales[0], p_factor=scales[1], e_factor=scales[2])
    got = ON.run(model, {"x": x.numpy(), "x_lengths": xl.numpy(), "scales": np.asarray(scales, dtype=np.float32)})
    assert got["durations"].dtype == np.int64 and np.array_equal(got["durations"], want["durations"].numpy())
    assert np.array_equal(got["wav_lengths"], want["wav_lengths"].numpy())
    w = want["wav"].detach().numpy()
    assert got["wav"].shape == w.shape
    assert np.abs(got["wav"] - w).max() <= 1e-4 * np.abs(w).max(), np.abs(got["wav"] - w).max() / np.abs(w).max()


def test_onnx_graph_vs_reference_golden(tmp_path, golden):
    """Against values the REFERENCE's synthesise produced (tests/golden/synth_small.npz)."""
    from optispeech_amd.onnx_export import export_as_onnx
    g = golden("synth_small")
    m, W = _small_model(int(g["seed"]), float(g["dur_bias"]))
    model = ON.parse_model(open(export_as_onnx(m, str(tmp_path / "m.onnx")), "rb").read())
    got = ON.run(model, {"x": g["in_x"], "x_lengths": g["in_x_lengths"], "scales": np.asarray([1.1, 1.6, 1.2], dtype=np.float32)})
    assert np.array_equal(got["durations"], g["durations"]) and np.array_equal(got["wav_lengths"], g["wav_lengths"])
    assert np.abs(got["wav"] - g["wav"]).max() <= 1e-3 * np.abs(g["wav"]).max()


def test_multispeaker_graph_takes_sids(tmp_path):
    from oracle import schema as S
    from optispeech_amd.config import make_optispeech
    from optispeech_amd.onnx_export import export_as_onnx
    from tests.test_gpu_generator import _small_cfg
    cfg = _small_cfg()
    cfg.num_speakers = 3
    m = make_optispeech(cfg, batch_size=2).eval()
    model = ON.parse_model(open(export_as_onnx(m, str(tmp_path / "m.onnx")), "rb").read())
    assert [n for n, _, _ in model["graph"]["inputs"]] == ["x", "x_lengths", "scales", "sids"]          # export.py:57-60
    x = torch.randint(1, 159, (2, 11), generator=torch.Generator().manual_seed(1)).numpy()
    feeds = {"x": x, "x_lengths": np.asarray([11, 11]), "scales": np.ones(3, dtype=np.float32)}
    a = ON.run(model, dict(feeds, sids=np.asarray([0, 0])))
    b = ON.run(model, dict(feeds, sids=np.asarray([0, 2])))
    assert np.array_equal(a["wav"][0], b["wav"][0]) or a["wav"].shape != b["wav"].shape or not np.array_equal(a["wav"][1], b["wav"][1])


@pytest.mark.gpu
def test_onnx_graph_reproduces_hip_synthesise(tmp_path):
    """The exported graph against the product's own synthesise() on the GPU (f32 mode): integer outputs exact, waveform 1e-3."""
    from optispeech_amd import precision
    from optispeech_amd.onnx_export import export_as_onnx
    from optispeech_amd.values import InferenceInputs
    precision.set_precision("f32")
    m, W = _small_model()
    m = m.to("cuda")
    model = ON.parse_model(open(export_as_onnx(m, str(tmp_path / "m.onnx")), "rb").read())
    xl = torch.tensor([21, 13, 30])
    Tt = int(xl.max())
    x = torch.randint(1, 159, (3, Tt), generator=torch.Generator().manual_seed(5)) * (torch.arange(Tt)[None] < xl[:, None])
    out = m.generator.synthesise(x.to("cuda"), xl, d_factor=1.1, p_factor=0.9, e_factor=1.2)
    got = ON.run(model, {"x": x.numpy(), "x_lengths": xl.numpy(), "scales": np.asarray([1.1, 0.9, 1.2], dtype=np.float32)})
    assert np.array_equal(got["durations"], out["durations"].numpy()) and np.array_equal(got["wav_lengths"], out["wav_lengths"].numpy())
    w = out["wav"].numpy()
    assert np.abs(got["wav"] - w).max() <= 1e-3 * np.abs(w).max()
