#!/usr/bin/env python3
"""Golden fixtures for the other backbones by RUNNING THE REFERENCE modules here.

    python tools/make_golden_backbones.py        # writes tests/golden/lightspeech_{enc,dec}.npz, leanspeech.npz, conformer.npz

optispeech.model.generator.modules.LightSpeechTransformerEncoder / ...Decoder (modules/lightspeech_transformer.py:14-96,
EncSepConvLayer / ConvSeparable modules/layers.py:455-506) are instantiated in eval mode (dropout off) at a reduced width with
the kernel sizes of configs/model/generator/{encoder,decoder}/lightspeech_transformer.yaml, fed a ragged batch; the output and
the gradients of sum(out * G) w.r.t. the input and every parameter are stored with the (small) state dict.  No reference source
is copied.
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
from optispeech.model.generator.modules.conformer import Conformer  # noqa: E402
from optispeech.model.generator.modules.leanspeech import LeanSpeechBackbone  # noqa: E402
from optispeech.model.generator.modules.lightspeech_transformer import (LightSpeechTransformerDecoder,  # noqa: E402
                                                                        LightSpeechTransformerEncoder)


def run(name, m, seed):
    torch.manual_seed(seed)
    with torch.no_grad():                           # move the parameters off their init pattern (biases are zero at init)
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)
        for n, b in m.named_buffers():              # BatchNorm running statistics off (0, 1): the eval path must use them
            if n.endswith("running_mean"):
                b.copy_(torch.randn_like(b) * 0.1)
            elif n.endswith("running_var"):
                b.copy_(torch.rand_like(b) + 0.5)
    B, T, C = 3, 41, 64
    lens = torch.tensor([41, 23, 6])
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
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
    print(name, "\n  " + "\n  ".join(f"{k} {tuple(v.shape)}" for k, v in m.state_dict().items()))


torch.manual_seed(21)
run("lightspeech_enc", LightSpeechTransformerEncoder(dim=64, kernel_sizes=[5, 25, 13, 9], activation="relu", dropout=0.2).eval(), 22)
torch.manual_seed(23)
run("lightspeech_dec", LightSpeechTransformerDecoder(dim=64, kernel_sizes=[17, 21, 9, 13], activation="relu", dropout=0.2,
                                                     max_source_positions=2000).eval(), 24)
torch.manual_seed(25)
# LeanSpeechBackbone (modules/leanspeech.py:14-97; configs/model/generator/encoder/leanspeech.yaml: kernel_size 9, drop_path 0.2), 2 layers
run("leanspeech", LeanSpeechBackbone(dim=64, kernel_size=9, num_layers=2, drop_path=0.2).eval(), 26)
torch.manual_seed(27)
# Conformer (modules/conformer.py; configs/model/generator/encoder/conformer.yaml at a reduced width, 2 blocks)
run("conformer", Conformer(dim=64, attention_heads=2, linear_units=96, num_blocks=2, dropout_rate=0.2, positional_dropout_rate=0.2,
                           attention_dropout_rate=0.2, normalize_before=True, concat_after=False, positionwise_layer_type="conv1d",
                           positionwise_conv_kernel_size=1, macaron_style=True, pos_enc_layer_type="rel_pos",
                           selfattention_layer_type="rel_selfattn", activation_type="swish", use_cnn_module=True, cnn_module_kernel=7,
                           zero_triu=False, init_type="xavier_uniform").eval(), 28)
