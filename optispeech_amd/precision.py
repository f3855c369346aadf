"""Compute-precision switch of the GEMM-shaped work.

"f32"  : exact-f32 MFMA everywhere (parity mode; what the 1e-3 reference-parity tests run).
"bf16" : bf16 MFMA operands with f32 accumulate for the discriminator stacks (and, where enabled, the generator
         GEMMs) -- BASELINE.json config[1] names bf16 as the training precision (the reference trains 16-mixed).
"""
_mode = {"v": "f32"}


def set_precision(mode: str):
    assert mode in ("f32", "bf16")
    _mode["v"] = mode


def get_precision() -> str:
    return _mode["v"]


def is_bf16() -> bool:
    return _mode["v"] == "bf16"
