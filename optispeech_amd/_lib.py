"""ctypes binding of libosp_hip.so -- the C-ABI boundary (include/osp.h).

There is NO fallback: if the shared object is missing or a call fails, we raise.  Tensors are passed
as raw device pointers, sizes as int64, real scalars as float, and the launch goes to torch's
current HIP stream, so kernels order with surrounding torch work and can be graph-captured.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libosp_hip.so")


class OspError(RuntimeError):
    pass


_CTYPE = {"int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double, "int": ctypes.c_int}


def _header_signatures():
    """argtypes of every entry point, parsed from include/osp.h (the header IS the boundary; it is generated from the
    sources and checked by tests/test_abi.py).  Pointers and hipStream_t -> void*."""
    path = os.path.join(os.path.dirname(_HERE), "include", "osp.h")
    sigs = {}
    if not os.path.exists(path):
        return sigs
    import re
    text = open(path).read()
    for m in re.finditer(r"^\s*int\s+(osp_\w+)\s*\(([^;]*)\)\s*;", text, flags=re.M):
        types = []
        for prm in m.group(2).split(","):
            prm = prm.strip()
            if not prm or prm == "void":
                continue
            if "*" in prm or prm.startswith("hipStream_t"):
                types.append(ctypes.c_void_p)
            else:
                types.append(_CTYPE[prm.replace("const ", "").split()[0]])
        sigs[m.group(1)] = types
    return sigs


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise OspError(f"{LIB_PATH} not found: build it with `python -m optispeech_amd.build` "
                           "(the HIP extension is mandatory; there is no CPU/eager fallback)")
        from . import build as _build
        if os.path.isdir(_build.CSRC) and _build.library_hash(LIB_PATH) != _build.source_hash():
            raise OspError(f"{LIB_PATH} was not built from the sources in {_build.CSRC} (content hash mismatch): rebuild it "
                           "with `python -m optispeech_amd.build`")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.cdll.osp_last_error.restype = ctypes.c_char_p
        self._fn = {}
        self._sigs = _header_signatures()
        # native argument marshalling (optispeech_amd/fastcall.py): same library, ~5x less interpreter time per call
        self._fcall, self._fidx, self._fast = None, {}, None
        self.ncalls = 0
        if os.environ.get("OSP_CTYPES_CALL", "0") != "1":
            fast = _load_fast()
            if fast is not None and fast.HEADER_SHA1 != _header_sha1():
                import warnings
                warnings.warn("optispeech_amd/lib/_ospfast*.so was generated from a different include/osp.h: ignoring it "
                              "(ctypes marshalling); rebuild with `python -m optispeech_amd.build`")
                fast = None
            if fast is not None:
                fast.set_guard(os.environ.get("OSP_FAST_CALL", "0") != "1")
                self._fast = fast
                self._fcall = fast.call
                self._fidx = {n: fast.index(n) for n in self._sigs if fast.index(n) is not None}

    def fn(self, name):
        f = self._fn.get(name)
        if f is None:
            f = getattr(self.cdll, name)
            f.restype = ctypes.c_int
            if name in self._sigs:
                f.argtypes = self._sigs[name]       # ctypes converts ints / floats / None itself: no per-argument wrappers
            self._fn[name] = f
        return f

    def call(self, name, *args):
        if _RECORD[0] is not None:                     # ops.side_wgrad: the launch is deferred to the next flush onto the side stream
            _RECORD[0].append((name, args))
            return
        # hot path (~850 calls per training step): keep the per-argument work minimal.  ctypes converts ints / floats /
        # None through the header-derived argtypes; tensors are passed as their device address.
        idx = self._fidx.get(name)
        dev = _cur_device()
        self.ncalls += 1                               # (direct calls; a tape's replayed calls are counted by the tape)
        if _GUARD:
            # kernels go to the current stream of torch's CURRENT device: a tensor living on another GPU would be addressed
            # from the wrong device's stream (fault, or silent peer access unordered with that GPU's work)
            for a in args:
                if isinstance(a, torch.Tensor):
                    if a.is_cuda and a.device.index != dev:
                        raise OspError(f"{name}: tensor on cuda:{a.device.index} but the current device is cuda:{dev}; "
                                       "call torch.cuda.set_device() (or wrap the call in torch.cuda.device(...))")
                    break
        if idx is not None:
            rc = self._fcall(idx, _STREAM_OVERRIDE[0] or _raw_stream(dev), *args)
            if rc != 0:
                raise OspError(f"{name} failed ({rc}): {self.cdll.osp_last_error().decode()}")
            return
        if self._fast is not None and hasattr(self._fast, "tape_recording") and self._fast.tape_recording():
            raise OspError(f"{name} has no native marshalling entry (include/osp.h changed?): it cannot be recorded on a tape")
        f = self._fn.get(name) or self.fn(name)
        T = torch.Tensor
        cargs = [a.data_ptr() if isinstance(a, T) else (a.ctypes.data if hasattr(a, "ctypes") else a) for a in args]
        if _GUARD and any(isinstance(a, T) and not a.is_cuda for a in args):
            raise OspError(f"{name}: tensor argument is not on the GPU")
        cargs.append(_STREAM_OVERRIDE[0] or _raw_stream(dev))      # torch's current HIP stream (raw handle)
        try:
            rc = f(*cargs)
        except ctypes.ArgumentError as e:
            raise TypeError(f"{name}: {e} (argument types must match include/osp.h)") from None
        except TypeError:
            if f.argtypes is None:
                raise OspError(f"{name} is not declared in include/osp.h (regenerate it with tools/gen_header.py)") from None
            raise TypeError(f"{name}: {len(cargs) - 1} arguments given, include/osp.h declares {len(f.argtypes) - 1}") from None
        if rc != 0:
            if any(isinstance(a, T) and not a.is_cuda for a in args):
                raise OspError(f"{name}: tensor argument is not on the GPU")
            raise OspError(f"{name} failed ({rc}): {self.cdll.osp_last_error().decode()}")


#: reject host tensors before launching (a host pointer would fault on the device); OSP_FAST_CALL=1 drops the check
_GUARD = os.environ.get("OSP_FAST_CALL", "0") != "1"
#: raw hipStream_t that replaces "torch's current stream" for the launches of a region (ops.side_wgrad: weight-gradient kernels on
#: a side stream without switching torch's current stream -- no torch op runs inside such a region); None = off
_STREAM_OVERRIDE = [None]
#: list that collects (entry point, arguments) instead of launching (ops.side_wgrad defers weight-gradient launches); None = off
_RECORD = [None]
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def _header_sha1():
    import hashlib
    path = os.path.join(os.path.dirname(_HERE), "include", "osp.h")
    return hashlib.sha1(open(path, "rb").read()).hexdigest() if os.path.exists(path) else ""


def _load_fast():
    """The _ospfast extension next to libosp_hip.so, or None when it has not been built (ctypes then does the marshalling)."""
    import glob
    import importlib.machinery
    import importlib.util
    hits = glob.glob(os.path.join(_HERE, "lib", "_ospfast*.so"))
    if not hits:
        return None
    loader = importlib.machinery.ExtensionFileLoader("_ospfast", hits[0])
    spec = importlib.util.spec_from_loader("_ospfast", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def call(name, *args):
    lib().call(name, *args)
