#!/usr/bin/env python3
"""Golden fixture for the Transformer backbone variant by RUNNING THE REFERENCE module here.

    python tools/make_golden_transformer.py        # writes tests/golden/transformer.npz

optispeech.model.generator.modules.Transformer (modules/transformer.py:9-27 -> _transformer/encoder.py) is instantiated in
eval mode (dropout off) at a reduced width, fed a ragged batch, and its output plus the gradients of sum(out * G) w.r.t.
the input and every parameter are stored together with the (small) state dict.  No reference source is copied.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from tools.make_golden import install_stubs  # noqa: E402

install_stubs()
from optispeech.model.generator.modules.transformer import Transformer  # noqa: E402

CFG = dict(attention_heads=2, linear_units=96, num_blocks=2, dropout_rate=0.2, positional_dropout_rate=0.2,
           attention_dropout_rate=0.2, normalize_before=True, concat_after=False, positionwise_layer_type="conv1d",
           positionwise_conv_kernel_size=1, use_scaled_pos_enc=True, init_alpha=1.0, init_type="xavier_uniform")
torch.manual_seed(11)
m = Transformer(dim=64, **CFG).eval()
with torch.no_grad():                               # move the parameters off their init pattern (biases are zero at init)
    for p in m.parameters():
        p.add_(torch.randn_like(p) * 0.05)
B, T, C = 3, 37, 64
lens = torch.tensor([37, 20, 5])
x = torch.randn(B, T, C, requires_grad=True)
pad = torch.arange(T)[None] >= lens[:, None]
y = m(x, pad)
G = torch.randn_like(y)
(y * G).sum().backward()
out = {"x": x.detach().numpy(), "lens": lens.numpy(), "y": y.detach().numpy(), "G": G.numpy(), "dx": x.grad.numpy()}
for k, v in m.state_dict().items():
    out["w_" + k] = v.numpy()
for k, p in m.named_parameters():
    out["g_" + k] = p.grad.numpy()
out["keys"] = np.array(list(m.state_dict().keys()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "transformer.npz"), **out)
print("\n".join(f"{k} {tuple(v.shape)}" for k, v in m.state_dict().items()))
