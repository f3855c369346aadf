"""Compute-precision switch of the GEMM-shaped work.

"f32"   : exact-f32 MFMA everywhere (parity mode; what the 1e-3 reference-parity tests run).
"bf16"  : bf16 MFMA operands with f32 accumulate for the discriminator stacks and the generator GEMMs -- BASELINE.json
          config[1] names bf16 as the training precision (the reference trains 16-mixed).
"mixed" : the PARITY mode at bench speed.  The generator (everything that produces mel / wav_hat, its backward, the spectral
          reconstruction losses) keeps f32 tensors and f32 accumulation; only the MPD / MRD discriminator conv stacks -- 90 % of the
          step's flops, and no part of the synthesised waveform -- run on the bf16 kernels.  Since late round 5 the generator's GEMMs
          OUTSIDE the index-critical path (below) take ``osp_conv_gemm_f32_split``: f32 operands, every element entering the matrix
          pipe as a (hi, lo) pair of bf16 numbers, three bf16 MFMAs per product (<= 1.1e-5 of |a b| per product; the exact pipe
          6e-8, plain bf16 operands 4e-3) -- the index-critical forward stays on the exact-f32 kernels, so durations / alignment
          indices are the f32 mode's bit for bit.  wav_hat and mel stay inside north_star's 1e-3 (tests/test_gpu_mixed.py,
          tests/test_gpu_fullsize_golden.py assert it against the reference goldens); bench.py reports the step as
          ``parity_mode_step``.  OSP_F32_SPLIT=0: every generator GEMM on the exact-f32 kernels, as "f32".
"""
_mode = {"v": "f32"}
_mixed = {"v": False}
_in_fwd = {"v": False}


def set_precision(mode: str):
    assert mode in ("f32", "bf16", "mixed")
    _mixed["v"] = mode == "mixed"
    _mode["v"] = "f32" if mode == "mixed" else mode


def get_precision() -> str:
    """The EFFECTIVE mode: what kernels and dtypes code running right now must choose (inside parity_forward() that is "f32",
    consistent with is_bf16())."""
    return "mixed" if (_mixed["v"] and _mode["v"] == "f32") else _mode["v"]


def configured_precision() -> str:
    """The mode the user configured, whatever region is executing (inside parity_forward() of the bf16 mode: "bf16"): what keys of
    tapes and graphs are built from (signature())."""
    if _in_fwd["v"]:
        return "bf16"
    return get_precision()


def is_bf16() -> bool:
    return _mode["v"] == "bf16"


def signature():
    """What a recorded call tape / captured hipGraph of generator code depends on besides shapes: the configured mode AND the
    switches that choose kernels inside it.  Keys built from this never replay a region recorded under other switches."""
    return (configured_precision(), _split["v"], _split_wgrad["v"], _split_bwd["v"], _fwd_parity["v"], _index_f32["v"])


# ---- index-critical path
# The monotonic alignment search, the durations it yields and the duration predictor's integer output at inference are
# DISCRETE functions of the encoder output / the attention log-probabilities: bf16 operand rounding upstream moves a few
# token boundaries (7 of 256 tokens on the reference-run full-size golden), after which nothing downstream is comparable with
# the f32 reference.  north_star asks for bit-exact alignment / length-regulator indexing, so in bf16 mode the forward of the
# text embedding -> encoder -> alignment module (and, in synthesise, the duration predictor) stays on the exact-f32 kernels:
# ~75 GFLOP of the step's ~8 TFLOP (+0.9 ms at the BASELINE shape).  The same kernels on the same inputs as the f32 mode,
# hence bit-identical indices by construction.  OSP_INDEX_PATH_F32=0 switches it off (pure bf16, as autocast would run it).
import contextlib as _contextlib
import os as _os


@_contextlib.contextmanager
def disc_scope():
    """Around the discriminator stacks' FORWARD code ("mixed" only): their Functions pick the bf16 kernels here and keep
    that choice for their backward (disc_ops.ConvStackFn launches bf16 kernels explicitly), while every generator-side
    Function -- whose backward consults the mode again -- keeps seeing "f32"."""
    if _mixed["v"] and _mode["v"] == "f32":
        _mode["v"] = "bf16"
        try:
            yield
        finally:
            _mode["v"] = "f32"
    else:
        yield


@_contextlib.contextmanager
def generator_scope():
    """Inside a disc_scope: back to the generator-side precision (the spectral reconstruction losses live in the
    discriminator module but are part of the generator's objective and of its parity contract)."""
    if _mixed["v"] and _mode["v"] == "bf16":
        _mode["v"] = "f32"
        try:
            yield
        finally:
            _mode["v"] = "bf16"
    else:
        yield

_index_f32 = {"v": _os.environ.get("OSP_INDEX_PATH_F32", "1") != "0"}


def set_index_path_f32(on: bool):
    _index_f32["v"] = bool(on)


_entered = {"v": False}


#: inside index_path() in ANY mode (the "mixed" mode's split-bf16 GEMMs must not run there)
_in_index = {"v": False}
_split = {"v": _os.environ.get("OSP_F32_SPLIT", "1") != "0"}


def set_f32_split(on: bool):
    _split["v"] = bool(on)


_split_wgrad = {"v": _os.environ.get("OSP_F32_SPLIT_WGRAD", "1") != "0"}
_split_bwd = {"v": _os.environ.get("OSP_F32_SPLIT_BWD", "1") != "0"}


def f32_split(kind: str = "gemm") -> bool:
    """True where an f32 GEMM (kind "gemm": forward and input gradients) or weight gradient ("wgrad") may take the split-bf16
    kernel: "mixed" mode, generator side, outside the index-critical path.  OSP_F32_SPLIT_WGRAD=0 keeps the weight gradients,
    OSP_F32_SPLIT_BWD=0 everything inside an autograd backward pass, on the exact-f32 kernels."""
    if not (_split["v"] and (_mixed["v"] or _in_fwd["v"]) and _mode["v"] == "f32" and not _in_index["v"]):
        return False
    if kind == "wgrad" and not _split_wgrad["v"]:
        return False
    if not _split_bwd["v"]:
        import torch
        in_backward = getattr(torch._C, "_current_graph_task_id", None)      # >= 0 while an autograd backward pass runs
        if in_backward is not None and in_backward() >= 0:
            return False
    return True


@_contextlib.contextmanager
def index_path():
    was = _in_index["v"]
    _in_index["v"] = True
    try:
        if _mode["v"] == "bf16" and _index_f32["v"]:
            _mode["v"], _entered["v"] = "f32", True
            try:
                yield
            finally:
                _mode["v"], _entered["v"] = "bf16", False
        else:
            yield
    finally:
        _in_index["v"] = was


def leave_index_path():
    """Inside ``index_path()``: switch back to the configured precision for the rest of the block (the context's exit is then a
    no-op for the mode)."""
    _in_index["v"] = False
    if _entered["v"]:
        _mode["v"] = "bf16"


# ---- parity-grade FORWARD inside the bf16 mode (opt-in: OSP_FWD_PARITY=1 / set_forward_parity)
# The generator's forward of a training step runs beside the previous step's discriminator phase with ~1.5 ms to spare
# (profiles/r05_phase_events_final.txt), so it can afford the parity mode's kernels: inside this scope the bf16 mode behaves as
# "mixed" does for the generator -- f32 tensors, the index-critical part on the exact kernels, every other GEMM on split-bf16
# products -- while every backward pass (run outside the scope) and the discriminator stacks stay on the bf16 kernels.  What the
# step OUTPUTS (mel-side predictions, wav_hat, the acoustic losses) then carries the parity mode's error (~1e-5) instead of the bf16
# operands' (~6e-3); gradients are the bf16 mode's.  ConvNeXtBlockFn / the conv Functions decide their backward precision at
# backward time and take f32 saved activations in bf16 mode -- the combination the index-critical encoder blocks have always run.
_fwd_parity = {"v": _os.environ.get("OSP_FWD_PARITY", "0") == "1"}


def set_forward_parity(on: bool):
    _fwd_parity["v"] = bool(on)


@_contextlib.contextmanager
def parity_forward():
    if _fwd_parity["v"] and _mode["v"] == "bf16" and not _mixed["v"] and not _in_fwd["v"]:
        _mode["v"], _in_fwd["v"] = "f32", True
        try:
            yield
        finally:
            _mode["v"], _in_fwd["v"] = "bf16", False
    else:
        yield

