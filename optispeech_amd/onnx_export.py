"""``.onnx`` writer for ``synthesise`` (reference: optispeech/onnx/export.py:20-125, consumer onnx/infer.py:24-145).

The reference traces ``generator.synthesise`` with ``torch.onnx.export`` and stores the text-processor description as the
``inference`` metadata entry.  Neither ``onnx`` nor ``onnxruntime`` exists in this image, so the file is written directly:

  * a ~60-line protobuf encoder for the handful of ONNX messages a model file consists of (field numbers of onnx.proto3);
  * a small graph builder that states ``synthesise`` (generator/__init__.py:194-301) in standard ONNX operators, opset 16 (the
    reference's DEFAULT_OPSET: LayerNorm is spelled out with ReduceMean, GELU with Erf);
  * the weights come from ``state_dict()``, i.e. in the reference's own schema and layouts.

Same contract as the reference's exported graph: inputs ``x`` int64 (batch, time), ``x_lengths`` int64 (batch), ``scales`` float32
(3) = (d_factor, p_factor, e_factor) [+ ``sids`` / ``lids`` int64 (batch) for multi-speaker / multi-language models]; outputs ``wav``
float32 (batch, frames * hop), ``wav_lengths`` int64 (batch), ``durations`` int64 (batch, time); metadata ``inference`` = the JSON
``OptiSpeechONNXModel.from_onnx_session`` reads.  ConvNeXt encoder / decoder + WaveNeXt vocoder (configs/model/optispeech.yaml).
The graph is checked by evaluating the written FILE with an independent numpy interpreter of the ONNX operator semantics
(tests/_onnx_numpy.py) against ``synthesise`` itself.
"""
import json
import math
import struct

import numpy as np

# ---------------------------------------------------------------------------------------------- protobuf (wire format)
FLOAT, UINT8, INT8, INT32, INT64, BOOL = 1, 2, 3, 6, 7, 9
_NP2ONNX = {np.dtype("float32"): FLOAT, np.dtype("int64"): INT64, np.dtype("int32"): INT32, np.dtype("bool"): BOOL,
            np.dtype("uint8"): UINT8, np.dtype("int8"): INT8}
A_FLOAT, A_INT, A_STRING, A_TENSOR, A_FLOATS, A_INTS = 1, 2, 3, 4, 6, 7


def _varint(v):
    v &= (1 << 64) - 1                                   # negative int64 -> 10-byte two's complement
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def f_varint(field, v):
    return _key(field, 0) + _varint(int(v))


def f_bytes(field, b):
    if isinstance(b, str):
        b = b.encode("utf-8")
    return _key(field, 2) + _varint(len(b)) + bytes(b)


def f_float(field, v):
    return _key(field, 5) + struct.pack("<f", float(v))


def f_packed_ints(field, vals):
    return f_bytes(field, b"".join(_varint(int(v)) for v in vals))


def tensor_proto(name, arr):
    arr = np.asarray(arr).copy(order="C")                # (np.ascontiguousarray would turn a 0-d tensor into shape (1,))
    dt = _NP2ONNX[arr.dtype]
    body = f_packed_ints(1, arr.shape) if arr.ndim else b""
    body += f_varint(2, dt) + f_bytes(8, name) + f_bytes(9, arr.tobytes())          # dims, data_type, name, raw_data (little endian)
    return body


def _attr(name, v):
    body = f_bytes(1, name)
    if isinstance(v, float):
        body += f_float(2, v) + f_varint(20, A_FLOAT)
    elif isinstance(v, (bool, int, np.integer)):
        body += f_varint(3, int(v)) + f_varint(20, A_INT)
    elif isinstance(v, str):
        body += f_bytes(4, v) + f_varint(20, A_STRING)
    elif isinstance(v, np.ndarray):
        body += f_bytes(5, tensor_proto("", v)) + f_varint(20, A_TENSOR)
    elif isinstance(v, (list, tuple)) and all(isinstance(e, (int, np.integer)) for e in v):
        body += f_packed_ints(8, v) + f_varint(20, A_INTS)
    elif isinstance(v, (list, tuple)):
        body += f_bytes(7, b"".join(struct.pack("<f", float(e)) for e in v)) + f_varint(20, A_FLOATS)
    else:
        raise TypeError(f"attribute {name}: {type(v)}")
    return body


def _value_info(name, elem_type, dims):
    shape = b""
    for d in dims:
        shape += f_bytes(1, f_bytes(2, d) if isinstance(d, str) else f_varint(1, d))     # Dimension: dim_param | dim_value
    ttype = f_varint(1, elem_type) + f_bytes(2, shape)                                  # TypeProto.Tensor: elem_type, shape
    return f_bytes(1, name) + f_bytes(2, f_bytes(1, ttype))                             # ValueInfoProto: name, type{tensor_type}


# ---------------------------------------------------------------------------------------------- graph builder
class Graph:
    def __init__(self, name):
        self.name, self.nodes, self.inits, self.inputs, self.outputs = name, [], [], [], []
        self._n = 0
        self._consts = {}

    def _fresh(self, hint):
        self._n += 1
        return f"{hint}_{self._n}"

    def input(self, name, elem_type, dims):
        self.inputs.append(_value_info(name, elem_type, dims))
        return name

    def output(self, value, name, elem_type, dims):
        self.nodes.append(f_bytes(1, value) + f_bytes(2, name) + f_bytes(3, self._fresh("out")) + f_bytes(4, "Identity"))
        self.outputs.append(_value_info(name, elem_type, dims))

    def init(self, name, arr):
        self.inits.append(tensor_proto(name, arr))
        return name

    def const(self, arr, hint="c"):
        arr = np.asarray(arr)
        key = (arr.dtype.str, arr.shape, arr.tobytes())
        if key not in self._consts:
            self._consts[key] = self.init(self._fresh(hint), arr)
        return self._consts[key]

    def i64(self, *v):
        return self.const(np.asarray(v, dtype=np.int64))

    def f32(self, v):
        return self.const(np.asarray(v, dtype=np.float32))

    def op(self, op_type, inputs, n_out=1, **attrs):
        outs = [self._fresh(op_type.lower()) for _ in range(n_out)]
        body = b"".join(f_bytes(1, i) for i in inputs) + b"".join(f_bytes(2, o) for o in outs)
        body += f_bytes(3, self._fresh("n")) + f_bytes(4, op_type)
        body += b"".join(f_bytes(5, _attr(k, v)) for k, v in attrs.items())
        self.nodes.append(body)
        return outs[0] if n_out == 1 else outs

    def serialize(self):
        return (b"".join(f_bytes(1, n) for n in self.nodes) + f_bytes(2, self.name) + b"".join(f_bytes(5, t) for t in self.inits)
                + b"".join(f_bytes(11, i) for i in self.inputs) + b"".join(f_bytes(12, o) for o in self.outputs))


def model_proto(graph, opset, metadata):
    body = f_varint(1, 8)                                                               # ir_version 8 (opset 16)
    body += f_bytes(2, "optispeech_amd") + f_bytes(3, "3") + f_bytes(7, graph.serialize())
    body += f_bytes(8, f_bytes(1, "") + f_varint(2, opset))                             # opset_import {domain "", version}
    for k, v in metadata.items():
        body += f_bytes(14, f_bytes(1, k) + f_bytes(2, v))                              # metadata_props
    return body


# ---------------------------------------------------------------------------------------------- the synthesise graph
class _Syn:
    """Activations are channels-last (B, T, C) float32, as everywhere in this package; Conv nodes get an NCW view."""

    def __init__(self, g, W):
        self.g, self.W, self._made = g, W, {}

    def w(self, key, arr=None):
        """Initializer of a state-dict entry (under its own key) or of a re-laid-out copy of it (fresh name); one per key."""
        ck = (key, arr is None)
        if ck not in self._made:
            a = self.W[key] if arr is None else arr
            self._made[ck] = self.g.init(key if arr is None else self.g._fresh(key.replace(".", "_")), np.asarray(a, dtype=np.float32).copy(order="C"))
        return self._made[ck]

    # -- elementary pieces
    def layer_norm(self, x, pre, eps):
        g = self.g
        mean = g.op("ReduceMean", [x], axes=[-1], keepdims=1)
        xc = g.op("Sub", [x, mean])
        var = g.op("ReduceMean", [g.op("Mul", [xc, xc])], axes=[-1], keepdims=1)
        y = g.op("Div", [xc, g.op("Sqrt", [g.op("Add", [var, g.f32(eps)])])])
        return g.op("Add", [g.op("Mul", [y, self.w(pre + "weight")]), self.w(pre + "bias")])

    def conv_cl(self, x, wkey, bkey, pad, group=1):
        g = self.g
        k = self.W[wkey].shape[-1]
        ins = [g.op("Transpose", [x], perm=[0, 2, 1]), self.w(wkey)] + ([self.w(bkey)] if bkey else [])
        y = g.op("Conv", ins, kernel_shape=[int(k)], pads=[int(pad), int(pad)], strides=[1], dilations=[1], group=int(group))
        return g.op("Transpose", [y], perm=[0, 2, 1])

    def linear(self, x, wkey, bkey=None):
        y = self.g.op("MatMul", [x, self.w(wkey, self.W[wkey].T)])                      # (.., in) @ (in, out)
        return self.g.op("Add", [y, self.w(bkey)]) if bkey else y

    def gelu(self, x):
        g = self.g
        e = g.op("Erf", [g.op("Mul", [x, g.f32(1.0 / math.sqrt(2.0))])])
        return g.op("Mul", [g.op("Mul", [x, g.f32(0.5)]), g.op("Add", [e, g.f32(1.0)])])

    # -- modules
    def convnext_backbone(self, x, pre, keep3):
        g, i = self.g, 0
        while (pre + f"convnext.{i}.gamma") in self.W:
            p = pre + f"convnext.{i}."
            C = self.W[p + "gamma"].shape[0]
            h = self.conv_cl(x, p + "dwconv.weight", p + "dwconv.bias", 3, group=C)     # convnext.py:36
            h = self.layer_norm(h, p + "norm.", 1e-6)                                   # :38
            h = self.gelu(self.linear(h, p + "pwconv1.weight", p + "pwconv1.bias"))     # :39-40
            h = self.linear(h, p + "pwconv2.weight", p + "pwconv2.bias")                # :41
            h = g.op("Mul", [h, self.w(p + "gamma")])                                   # :42-43
            x = g.op("Mul", [g.op("Add", [x, h]), keep3])                               # :46, :99-101
            i += 1
        return self.layer_norm(x, pre + "final_layer_norm.", 1e-6)

    def variance_predictor(self, x, valid, pre):
        g, i = self.g, 0
        while (pre + f"conv.{i}.0.weight") in self.W:
            k = self.W[pre + f"conv.{i}.0.weight"].shape[-1]
            x = g.op("Relu", [self.conv_cl(x, pre + f"conv.{i}.0.weight", pre + f"conv.{i}.0.bias", (k - 1) // 2)])
            x = self.layer_norm(x, pre + f"conv.{i}.2.", 1e-12)
            i += 1
        y = g.op("Squeeze", [self.linear(x, pre + "linear.weight", pre + "linear.bias"), g.i64(-1)])
        return g.op("Where", [valid, y, g.f32(0.0)])                                    # core.py:96

    def variance_embed_add(self, x, values, keep3, pre):
        g = self.g
        k = self.W[pre + "embed.0.weight"].shape[-1]
        emb = self.conv_cl(g.op("Unsqueeze", [values, g.i64(2)]), pre + "embed.0.weight", pre + "embed.0.bias", (k - 1) // 2)
        return g.op("Mul", [g.op("Add", [x, emb]), keep3])


def _length_mask(g, lengths, T):
    """(B, T) bool, True where t < length (utils/model.py:12-16); T: int64 scalar tensor name."""
    ar = g.op("Range", [g.const(np.asarray(0, dtype=np.int64)), T, g.const(np.asarray(1, dtype=np.int64))])
    return g.op("Less", [g.op("Unsqueeze", [ar, g.i64(0)]), g.op("Unsqueeze", [lengths, g.i64(1)])]), ar


def build_synthesise_graph(W, *, hop_length, num_speakers=1, num_languages=1, theta=2000.0, delta=0.1, clip_val=1e-8):
    """W: {reference state-dict key under ``generator.``: numpy array}.  Returns the Graph."""
    g = Graph("optispeech_synthesise")
    pre = "generator."
    s = _Syn(g, W)
    x = g.input("x", INT64, ["batch_size", "time"])
    xl = g.input("x_lengths", INT64, ["batch_size"])
    scales = g.input("scales", FLOAT, [3])
    d_factor, p_factor, e_factor = (g.op("Gather", [scales, g.const(np.asarray(i, dtype=np.int64))], axis=0) for i in range(3))
    Tt = g.op("Gather", [g.op("Shape", [x]), g.const(np.asarray(1, dtype=np.int64))], axis=0)
    valid, ar = _length_mask(g, xl, Tt)                                                  # generator/__init__.py:224-226
    keep3 = g.op("Unsqueeze", [g.op("Cast", [valid], to=FLOAT), g.i64(2)])
    # TextEmbedding (modules/core.py:25-31, layers.py:48-71)
    dim = W[pre + "text_embedding.embed_tokens.weight"].shape[1]
    emb = g.op("Mul", [g.op("Gather", [s.w(pre + "text_embedding.embed_tokens.weight"), x], axis=0), g.f32(math.sqrt(dim))])
    half = dim // 2
    inv_freq = (theta ** -(np.arange(half, dtype=np.float32) / half)).astype(np.float32)
    ang = g.op("Mul", [g.op("Unsqueeze", [g.op("Cast", [ar], to=FLOAT), g.i64(1)]), g.const(inv_freq[None, :])])
    pos = g.op("Mul", [g.op("Concat", [g.op("Sin", [ang]), g.op("Cos", [ang])], axis=-1), s.w(pre + "text_embedding.embed_positions.scale")])
    h = g.op("Add", [emb, g.op("Unsqueeze", [pos, g.i64(0)])])
    h = s.convnext_backbone(h, pre + "encoder.", keep3)                                  # :232
    if num_speakers > 1:
        sids = g.input("sids", INT64, ["batch_size"])
        h = g.op("Add", [h, g.op("Unsqueeze", [g.op("Gather", [s.w(pre + "sid_embed.weight"), sids], axis=0), g.i64(1)])])
    if num_languages > 1:
        lids = g.input("lids", INT64, ["batch_size"])
        h = g.op("Add", [h, g.op("Unsqueeze", [g.op("Gather", [s.w(pre + "lid_embed.weight"), lids], axis=0), g.i64(1)])])
    # DurationPredictor.infer (core.py:118-132): ceil((exp(log d) - clip) * factor), >= 0, 0 at padding
    log_d = s.variance_predictor(h, valid, pre + "duration_predictor.")
    d = g.op("Ceil", [g.op("Mul", [g.op("Sub", [g.op("Exp", [log_d]), g.f32(clip_val)]), d_factor])])
    dur = g.op("Max", [g.op("Cast", [d], to=INT64), g.const(np.asarray(0, dtype=np.int64))])
    dur = g.op("Where", [valid, dur, g.const(np.asarray(0, dtype=np.int64))])
    # pitch / energy (core.py:168-175)
    pitch = g.op("Mul", [s.variance_predictor(h, valid, pre + "pitch_predictor.predictor."), p_factor])
    h = s.variance_embed_add(h, pitch, keep3, pre + "pitch_predictor.")
    energy = g.op("Mul", [s.variance_predictor(h, valid, pre + "energy_predictor.predictor."), e_factor])
    h = s.variance_embed_add(h, energy, keep3, pre + "energy_predictor.")
    # all-zero durations -> ones (alignments.py:152-157)
    total = g.op("ReduceSum", [dur], keepdims=0)
    dur = g.op("Where", [g.op("Equal", [total, g.const(np.asarray(0, dtype=np.int64))]),
                         g.op("Add", [g.op("Mul", [dur, g.const(np.asarray(0, dtype=np.int64))]), g.const(np.asarray(1, dtype=np.int64))]), dur])
    y_len = g.op("ReduceSum", [dur, g.i64(1)], keepdims=0)                                # :258
    Tm = g.op("ReduceMax", [y_len], keepdims=0)
    y_valid, ar_y = _length_mask(g, y_len, Tm)
    ykeep = g.op("Cast", [y_valid], to=FLOAT)
    # GaussianUpsampling (alignments.py:136-174)
    ds = g.op("Cast", [dur], to=FLOAT)
    t = g.op("Mul", [g.op("Unsqueeze", [g.op("Cast", [ar_y], to=FLOAT), g.i64(0)]), ykeep])                       # :163-165
    c = g.op("Sub", [g.op("CumSum", [ds, g.const(np.asarray(1, dtype=np.int64))]), g.op("Mul", [ds, g.f32(0.5)])])   # :167
    diff = g.op("Sub", [g.op("Unsqueeze", [t, g.i64(2)]), g.op("Unsqueeze", [c, g.i64(1)])])
    en = g.op("Mul", [g.op("Mul", [diff, diff]), g.f32(-delta)])                                                      # :168
    en = g.op("Where", [g.op("Unsqueeze", [valid, g.i64(1)]), en, g.f32(-np.inf)])                                    # :170
    y = g.op("MatMul", [g.op("Softmax", [en], axis=2), h])                                                            # :172-173
    ykeep3 = g.op("Unsqueeze", [ykeep, g.i64(2)])
    y = s.convnext_backbone(y, pre + "decoder.", ykeep3)                                  # :268
    # WaveNeXt (vocoder/wavenext/__init__.py:31-48, 82-86)
    v = pre + "vocoder."
    z = s.conv_cl(y, v + "embed.weight", v + "embed.bias", 3)
    z = s.layer_norm(z, v + "norm.", 1e-6)
    z = s.convnext_backbone(z, v + "backbone.", ykeep3)
    z = s.linear(s.linear(z, v + "head.linear_1.weight", v + "head.linear_1.bias"), v + "head.linear_2.weight")
    wav = g.op("Reshape", [z, g.i64(0, -1)])                                              # (B, frames * hop)
    wav = g.op("Clip", [wav, g.f32(-1.0), g.f32(1.0)])
    g.output(wav, "wav", FLOAT, ["batch_size", "frames"])
    g.output(g.op("Mul", [y_len, g.const(np.asarray(hop_length, dtype=np.int64))]), "wav_lengths", INT64, ["batch_size"])
    g.output(dur, "durations", INT64, ["batch_size", "time"])
    return g


def inference_metadata(model):
    """The ``inference`` metadata entry (export.py:96-125): what OptiSpeechONNXModel.from_onnx_session parses."""
    tp = model.text_processor
    to_dict = getattr(tp, "asdict", None)
    tok = getattr(tp, "tokenizer", None)
    ia = model.inference_args
    return json.dumps(dict(
        name=getattr(model.data_args, "name", "optispeech"), sample_rate=int(model.sample_rate),
        inference_args=dict(d_factor=float(ia.d_factor), p_factor=float(ia.p_factor), e_factor=float(ia.e_factor)),
        input_symbols=list(getattr(tok, "input_symbols", []) or []), special_symbols=dict(getattr(tok, "special_symbols", {}) or {}),
        speakers=list(getattr(model, "speakers", []) or []), languages=list(getattr(tp, "languages", []) or []),
        unicode_norm_form="NFC", text_processor=(to_dict() if callable(to_dict) else {})))


def export_as_onnx(model, out_filename, opset=16):
    """Write ``synthesise`` of an ``OptiSpeech`` model to ``out_filename`` (export.py:20-93 + add_inference_metadata :96-125)."""
    import os
    gen = model.generator
    sd = {("generator." + k): v.detach().cpu().numpy() for k, v in gen.state_dict().items()}
    if "generator.encoder.convnext.0.gamma" not in sd or "generator.vocoder.backbone.convnext.0.gamma" not in sd:
        raise NotImplementedError("the .onnx writer covers the ConvNeXt encoder / decoder + WaveNeXt vocoder (configs/model/optispeech.yaml)")
    if opset < 13:
        raise ValueError("opset >= 13 (Squeeze / Unsqueeze / ReduceSum take their axes as inputs)")
    g = build_synthesise_graph(sd, hop_length=int(model.hop_length), num_speakers=int(gen.num_speakers), num_languages=int(gen.num_languages))
    blob = model_proto(g, opset, {"inference": inference_metadata(model)})
    os.makedirs(os.path.dirname(os.path.abspath(out_filename)), exist_ok=True)
    with open(out_filename, "wb") as fh:
        fh.write(blob)
    return out_filename
