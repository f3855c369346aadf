"""Test infrastructure: read an ``.onnx`` file back (generic protobuf wire decoding, onnx.proto3 field numbers) and evaluate its
graph with numpy, operator by operator, following the ONNX operator specifications (opset 13-17 forms) -- NOT the builder's
own helper functions.  ``onnxruntime`` is absent from the image; this is what stands in for it in tests/test_onnx_export.py."""
import struct

import numpy as np


# ------------------------------------------------------------------------------------------------ protobuf reader
def _read_varint(b, i):
    v, shift = 0, 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << shift
        if not c & 0x80:
            return v, i
        shift += 7


def fields(b):
    """[(field number, wire type, value)] of one message; length-delimited values stay bytes."""
    out, i = [], 0
    while i < len(b):
        key, i = _read_varint(b, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 2:
            n, i = _read_varint(b, i)
            v = b[i:i + n]; i += n
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError(f"wire type {wt}")
        out.append((fn, wt, v))
    return out


def _sint(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_ints(v):
    out, i = [], 0
    while i < len(v):
        x, i = _read_varint(v, i)
        out.append(_sint(x))
    return out


_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}


def parse_tensor(b):
    dims, dt, name, raw = [], 1, "", None
    for fn, wt, v in fields(b):
        if fn == 1:
            dims += _packed_ints(v) if wt == 2 else [_sint(v)]
        elif fn == 2:
            dt = v
        elif fn == 8:
            name = v.decode()
        elif fn == 9:
            raw = v
        elif fn in (4, 7, 5):
            raise NotImplementedError("typed data fields (the writer uses raw_data)")
    arr = np.frombuffer(raw, dtype=_DT[dt]).reshape(dims).copy()
    return name, arr


def parse_attr(b):
    name, val, typ = "", None, None
    ints, floats = [], []
    for fn, wt, v in fields(b):
        if fn == 1:
            name = v.decode()
        elif fn == 2:
            val = struct.unpack("<f", v)[0]
        elif fn == 3:
            val = _sint(v)
        elif fn == 4:
            val = v.decode()
        elif fn == 5:
            val = parse_tensor(v)[1]
        elif fn == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fn == 8:
            ints += _packed_ints(v) if wt == 2 else [_sint(v)]
        elif fn == 20:
            typ = v
    if typ == 7:
        val = ints
    elif typ == 6:
        val = floats
    return name, val


def parse_value_info(b):
    name, elem, dims = "", None, []
    for fn, wt, v in fields(b):
        if fn == 1:
            name = v.decode()
        elif fn == 2:
            for f2, _, v2 in fields(v):
                if f2 == 1:                                   # tensor_type
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:
                            elem = v3
                        elif f3 == 2:
                            for f4, _, v4 in fields(v3):      # dim
                                d = None
                                for f5, _, v5 in fields(v4):
                                    d = _sint(v5) if f5 == 1 else v5.decode()
                                dims.append(d)
    return name, elem, dims


def parse_model(blob):
    m = dict(ir_version=None, opset={}, metadata={}, graph=None, producer=None)
    for fn, wt, v in fields(blob):
        if fn == 1:
            m["ir_version"] = v
        elif fn == 2:
            m["producer"] = v.decode()
        elif fn == 7:
            g = dict(nodes=[], inits={}, inputs=[], outputs=[], name="")
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    node = dict(inputs=[], outputs=[], op=None, attrs={}, name="")
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:
                            node["inputs"].append(v3.decode())
                        elif f3 == 2:
                            node["outputs"].append(v3.decode())
                        elif f3 == 3:
                            node["name"] = v3.decode()
                        elif f3 == 4:
                            node["op"] = v3.decode()
                        elif f3 == 5:
                            k, val = parse_attr(v3)
                            node["attrs"][k] = val
                    g["nodes"].append(node)
                elif f2 == 2:
                    g["name"] = v2.decode()
                elif f2 == 5:
                    name, arr = parse_tensor(v2)
                    assert name not in g["inits"], f"duplicate initializer {name}"
                    g["inits"][name] = arr
                elif f2 == 11:
                    g["inputs"].append(parse_value_info(v2))
                elif f2 == 12:
                    g["outputs"].append(parse_value_info(v2))
            m["graph"] = g
        elif fn == 8:
            dom, ver = "", None
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    dom = v2.decode()
                elif f2 == 2:
                    ver = v2
            m["opset"][dom] = ver
        elif fn == 14:
            k = val = ""
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    k = v2.decode()
                elif f2 == 2:
                    val = v2.decode()
            m["metadata"][k] = val
    return m


# ------------------------------------------------------------------------------------------------ operators (ONNX semantics)
def _conv(x, w, b, a):
    import torch
    import torch.nn.functional as F
    assert x.ndim == 3 and a.get("kernel_shape", [w.shape[-1]]) == [w.shape[-1]]
    pads = a.get("pads", [0, 0])
    assert pads[0] == pads[1], "symmetric pads only"
    y = F.conv1d(torch.from_numpy(x), torch.from_numpy(w), None if b is None else torch.from_numpy(b), stride=a.get("strides", [1])[0],
                 padding=pads[0], dilation=a.get("dilations", [1])[0], groups=a.get("group", 1))
    return y.numpy()


def _erf(x):
    import torch
    return torch.erf(torch.from_numpy(np.ascontiguousarray(x))).numpy()


def _softmax(x, axis):
    m = np.max(x, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def _axes(v):
    return tuple(int(i) for i in np.asarray(v).reshape(-1))


_CAST = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}


def run(model, feeds):
    """Evaluate the graph; returns {output name: array}."""
    g = model["graph"]
    env = dict(g["inits"])
    for name, elem, dims in g["inputs"]:
        assert name in feeds, f"missing input {name}"
        arr = np.asarray(feeds[name])
        assert arr.dtype == _DT[elem], (name, arr.dtype, elem)
        assert arr.ndim == len(dims), (name, arr.shape, dims)
        env[name] = arr
    for nd in g["nodes"]:
        op, a = nd["op"], nd["attrs"]
        x = [env[i] if i else None for i in nd["inputs"]]
        if op == "Identity":
            y = x[0]
        elif op in ("Add", "Sub", "Mul", "Div"):
            f = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply, "Div": np.divide}[op]
            assert x[0].dtype == x[1].dtype, (op, nd["name"], x[0].dtype, x[1].dtype)      # ONNX: no implicit type promotion
            y = f(x[0], x[1])
            if np.issubdtype(x[0].dtype, np.integer) and op == "Div":
                y = (x[0] // x[1]).astype(x[0].dtype)
            y = y.astype(x[0].dtype)
        elif op == "Max":
            assert x[0].dtype == x[1].dtype
            y = np.maximum(x[0], x[1])
        elif op in ("Less", "Equal"):
            assert x[0].dtype == x[1].dtype, (op, x[0].dtype, x[1].dtype)
            y = (x[0] < x[1]) if op == "Less" else (x[0] == x[1])
        elif op == "Where":
            assert x[0].dtype == np.bool_ and x[1].dtype == x[2].dtype, (nd["name"], x[0].dtype, x[1].dtype, x[2].dtype)
            y = np.where(x[0], x[1], x[2])
        elif op == "Cast":
            y = x[0].astype(_CAST[a["to"]])
        elif op == "Shape":
            y = np.asarray(x[0].shape, dtype=np.int64)
        elif op == "Gather":
            y = np.take(x[0], x[1], axis=a.get("axis", 0))
        elif op == "Range":
            assert all(v.ndim == 0 for v in x)
            y = np.arange(x[0], x[1], x[2], dtype=x[0].dtype)
        elif op == "Unsqueeze":
            y = x[0]
            axes = _axes(x[1])
            nd_out = y.ndim + len(axes)
            for ax in sorted(i % nd_out for i in axes):
                y = np.expand_dims(y, ax)
        elif op == "Squeeze":
            y = np.squeeze(x[0], axis=_axes(x[1]))
        elif op in ("Sin", "Cos", "Exp", "Sqrt", "Ceil"):
            y = getattr(np, op.lower())(x[0]).astype(x[0].dtype)
        elif op == "Erf":
            y = _erf(x[0])
        elif op == "Relu":
            y = np.maximum(x[0], 0)
        elif op == "Concat":
            y = np.concatenate(x, axis=a["axis"])
        elif op == "Transpose":
            y = np.transpose(x[0], a["perm"])
        elif op == "Conv":
            y = _conv(x[0], x[1], x[2] if len(x) > 2 else None, a)
        elif op == "ReduceMean":                                  # opset 13-17: axes is an ATTRIBUTE
            y = np.mean(x[0], axis=tuple(a["axes"]) if "axes" in a else None, keepdims=bool(a.get("keepdims", 1))).astype(x[0].dtype)
        elif op == "ReduceMax":
            y = np.max(x[0], axis=tuple(a["axes"]) if "axes" in a else None, keepdims=bool(a.get("keepdims", 1)))
        elif op == "ReduceSum":                                   # opset 13+: axes is an INPUT
            ax = _axes(x[1]) if len(x) > 1 and x[1] is not None else None
            y = np.sum(x[0], axis=ax, keepdims=bool(a.get("keepdims", 1))).astype(x[0].dtype)
        elif op == "CumSum":
            assert not a.get("exclusive", 0) and not a.get("reverse", 0)
            y = np.cumsum(x[0], axis=int(x[1])).astype(x[0].dtype)
        elif op == "MatMul":
            y = np.matmul(x[0], x[1])
        elif op == "Softmax":
            y = _softmax(x[0], a.get("axis", -1))
        elif op == "Reshape":
            shp = [int(v) for v in x[1]]
            shp = [x[0].shape[i] if v == 0 else v for i, v in enumerate(shp)]   # allowzero = 0: a 0 copies the input dimension
            y = x[0].reshape(shp)
        elif op == "Clip":
            y = np.clip(x[0], x[1], x[2])
        else:
            raise NotImplementedError(op)
        env[nd["outputs"][0]] = np.asarray(y)
    return {name: env[name] for name, _, _ in g["outputs"]}
