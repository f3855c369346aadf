"""Parameter storage in kernel-native layouts, reference schema at the state-dict boundary.

Kernels want tap-major / channel-contiguous weights; checkpoints use torch's (Cout, Cin, k).
``RefSchemaModule`` converts in ``state_dict()`` / ``load_state_dict()`` so that reference
checkpoints (SURVEY.md section 8b key schema) load and save unchanged.
"""
import torch
from torch import nn


def conv_to_native(w):      # (Cout, Cin, k) -> (Cout, k, Cin)
    return w.permute(0, 2, 1).contiguous()


def conv_to_ref(w):         # (Cout, k, Cin) -> (Cout, Cin, k)
    return w.permute(0, 2, 1).contiguous()


def dw_to_native(w):        # (C, 1, 7) -> (7, C)
    return w[:, 0, :].t().contiguous()


def dw_to_ref(w):           # (7, C) -> (C, 1, 7)
    return w.t().contiguous()[:, None, :]


class RefSchemaModule(nn.Module):
    #: native parameter/buffer name -> (reference key, to_native, to_ref)
    _ref_layout = {}

    def _ref(self, name):
        return self._ref_layout.get(name, (name, None, None))

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        items = list(self._parameters.items()) + [(k, v) for k, v in self._buffers.items()
                                                  if k not in self._non_persistent_buffers_set]
        for name, t in items:
            if t is None:
                continue
            key, _, to_ref = self._ref(name)
            v = t if keep_vars else t.detach()
            destination[prefix + key] = to_ref(v) if to_ref else v

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        items = list(self._parameters.items()) + [(k, v) for k, v in self._buffers.items()
                                                  if k not in self._non_persistent_buffers_set]
        known = set()
        for name, t in items:
            if t is None:
                continue
            key, to_native, _ = self._ref(name)
            known.add(prefix + key)
            if prefix + key not in state_dict:
                missing_keys.append(prefix + key)
                continue
            v = state_dict[prefix + key]
            v = to_native(v) if to_native else v
            if tuple(v.shape) != tuple(t.shape):
                error_msgs.append(f"size mismatch for {prefix + key}: got {tuple(v.shape)}, want {tuple(t.shape)}")
                continue
            with torch.no_grad():
                t.copy_(v)
        if strict:
            child_prefixes = tuple(prefix + c + "." for c in self._modules)
            for k in state_dict:
                if k.startswith(prefix) and k not in known and not k.startswith(child_prefixes):
                    unexpected_keys.append(k)
