"""API value types of the hot path (mirror of optispeech/values.py:23-111): InferenceInputs / InferenceOutputs
with the same fields, defaults and helper methods (as_torch / as_numpy / to / unbatched_wavs)."""
import dataclasses
from dataclasses import dataclass
from typing import Any, Optional

import numpy as np
import torch


def _pad_rows(seqs, value=0):
    n = max(len(s) for s in seqs)
    out = np.full((len(seqs), n), value, dtype=np.int64)
    for i, s in enumerate(seqs):
        out[i, : len(s)] = np.asarray(s, dtype=np.int64)
    return out


@dataclass
class _Container:
    def as_tuple(self):
        return dataclasses.astuple(self)

    def as_dict(self):
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}

    def _map(self, fn):
        return type(self)(**{k: fn(v) for k, v in self.as_dict().items()})

    def as_torch(self):
        return self._map(lambda v: torch.as_tensor(v) if isinstance(v, (np.ndarray, torch.Tensor)) else v)

    def as_numpy(self):
        return self._map(lambda v: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v)

    def to(self, device):
        return self._map(lambda v: v.to(device) if isinstance(v, torch.Tensor) else v)


@dataclass(kw_only=True)
class InferenceInputs(_Container):
    clean_text: Any
    x: Any
    x_lengths: Any
    sids: Optional[Any] = None
    lids: Optional[Any] = None
    d_factor: float = 1.0
    p_factor: float = 1.0
    e_factor: float = 1.0

    @classmethod
    def from_ids_and_lengths(cls, ids, lengths, **kwargs):
        return cls(x=_pad_rows(ids), x_lengths=np.asarray(lengths, dtype=np.int64), **kwargs).as_numpy()


@dataclass(kw_only=True)
class InferenceOutputs(_Container):
    wav: Any
    wav_lengths: Any
    latency: float
    rtf: float
    durations: Optional[Any] = None
    pitch: Optional[Any] = None
    energy: Optional[Any] = None
    am_rtf: Optional[float] = None
    v_rtf: Optional[float] = None

    def __iter__(self):
        return iter(self.unbatched_wavs())

    def unbatched_wavs(self):
        return [self.wav[i, : int(n)] for i, n in enumerate(self.wav_lengths)]


# ---------------------------------------------------------------------------------------------- parameter epoch
# The fused optimizer updates the flat parameter arena through a raw pointer, which does not bump torch's tensor version
# counters.  Caches of quantities derived from parameters (the packed weight-norm operands of the discriminators) are
# keyed on this epoch in addition to ``Tensor._version``.
_param_epoch = 0


def param_epoch() -> int:
    return _param_epoch


def bump_param_epoch() -> None:
    global _param_epoch
    _param_epoch += 1
