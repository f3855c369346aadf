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


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise OspError(f"{LIB_PATH} not found: build it with `python -m optispeech_amd.build` "
                           "(the HIP extension is mandatory; there is no CPU/eager fallback)")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.cdll.osp_last_error.restype = ctypes.c_char_p
        self._fn = {}

    def fn(self, name):
        f = self._fn.get(name)
        if f is None:
            f = getattr(self.cdll, name)
            f.restype = ctypes.c_int
            self._fn[name] = f
        return f

    def call(self, name, *args):
        cargs = []
        for a in args:
            if a is None:
                cargs.append(ctypes.c_void_p(0))
            elif isinstance(a, torch.Tensor):
                if not a.is_cuda:
                    raise OspError(f"{name}: tensor argument is not on the GPU")
                cargs.append(ctypes.c_void_p(a.data_ptr()))
            elif isinstance(a, bool):
                cargs.append(ctypes.c_int64(int(a)))
            elif isinstance(a, int):
                cargs.append(ctypes.c_int64(a))
            elif isinstance(a, float):
                cargs.append(ctypes.c_float(a))
            else:
                raise TypeError(f"{name}: unsupported argument type {type(a)}")
        cargs.append(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        rc = self.fn(name)(*cargs)
        if rc != 0:
            raise OspError(f"{name} failed ({rc}): {self.cdll.osp_last_error().decode()}")


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def call(name, *args):
    lib().call(name, *args)
