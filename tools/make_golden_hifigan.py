#!/usr/bin/env python3
"""Golden fixture for the causal HiFi-GAN generator path by RUNNING THE REFERENCE modules here.

    python tools/make_golden_hifigan.py            # writes tests/golden/hifigan_small.npz

The reference package optispeech.model.vocoder.streaming_hifigan cannot be imported (its __init__ needs a discriminator module
that is not in the tree), but its three layer files can: modules/conv_layer.py (CausalConv1d,
CausalConvTranspose1d), modules/residual_block.py, modules/multi_fusion.py are loaded by file path.  The generator's forward
(streaming_hifigan/__init__.py:141-161) is a seven-line composition of those modules and is driven from here:
input_conv -> [LeakyReLU -> upsample_i -> MRF_i] -> LeakyReLU(0.01) -> output_conv -> tanh, weight norm on every conv.
Stored: the (small) state dict, the input, the output, the gradients of sum(out * G) w.r.t. the input and every parameter, and the
chunked streaming `inference()` output.  No reference source is copied.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = "/root/reference/optispeech/model/vocoder/streaming_hifigan"
pkg = types.ModuleType("shg"); pkg.__path__ = [BASE]; sys.modules["shg"] = pkg
mods = types.ModuleType("shg.modules"); mods.__path__ = [BASE + "/modules"]; sys.modules["shg.modules"] = mods
L = {}
for name in ("conv_layer", "residual_block", "multi_fusion"):
    spec = importlib.util.spec_from_file_location("shg.modules." + name, f"{BASE}/modules/{name}.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules["shg.modules." + name] = m
    spec.loader.exec_module(m)
    L[name] = m

CFG = dict(in_channels=16, out_channels=1, channels=32, kernel_size=7, upsample_scales=(4, 2), upsample_kernel_sizes=(8, 4),
           resblock_kernel_sizes=(3, 7, 11), resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)])


class RefGenerator(nn.Module):
    def __init__(self, c):
        super().__init__()
        CL, MF = L["conv_layer"], L["multi_fusion"]
        self.input_conv = CL.CausalConv1d(c["in_channels"], c["channels"], c["kernel_size"], stride=1)
        self.upsamples, self.blocks = nn.ModuleList(), nn.ModuleList()
        for i, (s, k) in enumerate(zip(c["upsample_scales"], c["upsample_kernel_sizes"])):
            self.upsamples.append(CL.CausalConvTranspose1d(c["channels"] // 2 ** i, c["channels"] // 2 ** (i + 1), kernel_size=k, stride=s))
            self.blocks.append(MF.MultiReceptiveField(channels=c["channels"] // 2 ** (i + 1), resblock_kernel_sizes=c["resblock_kernel_sizes"],
                                                      resblock_dilations=c["resblock_dilations"]))
        self.output_conv = CL.CausalConv1d(c["channels"] // 2 ** len(c["upsample_scales"]), c["out_channels"], c["kernel_size"], stride=1)
        self.act_up, self.act_out = nn.LeakyReLU(negative_slope=0.1), nn.LeakyReLU()
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d)):
                nn.utils.weight_norm(m)

    def forward(self, c, streaming=False):
        f = (lambda m, x: m.inference(x)) if streaming else (lambda m, x: m(x))
        c = f(self.input_conv, c)
        for up, blk in zip(self.upsamples, self.blocks):
            c = f(blk, f(up, self.act_up(c)))
        return torch.tanh(f(self.output_conv, self.act_out(c)))


def main():
    torch.manual_seed(21)
    g = RefGenerator(CFG)
    with torch.no_grad():                                        # O(1) weights so that every layer matters in the output
        for n, p in g.named_parameters():
            if n.endswith("weight_v"):
                p.copy_(torch.randn_like(p) / (p[0].numel() ** 0.5))
            elif n.endswith("weight_g"):
                p.copy_(0.7 + 0.3 * torch.rand_like(p))
            else:
                p.copy_(0.1 * torch.randn_like(p))
    B, T = 2, 23
    x = torch.randn(B, CFG["in_channels"], T, requires_grad=True)
    y = g(x)
    G = torch.randn_like(y)
    (y * G).sum().backward()
    out = {"x": x.detach().numpy(), "y": y.detach().numpy(), "G": G.numpy(), "dx": x.grad.numpy()}
    sd = {k: v for k, v in g.state_dict().items()}
    for k, v in sd.items():
        out["w_" + k] = v.numpy()
    for k, p in g.named_parameters():
        out["g_" + k] = p.grad.numpy()
    out["keys"] = np.array(list(sd.keys()))
    # streaming: one utterance (the reference's buffers are batch 1), chunks of 5 / 7 / 11 frames
    with torch.no_grad():
        x1 = x[:1].detach()
        chunks, pos = [], 0
        for n in (5, 7, 11):
            chunks.append(g(x1[:, :, pos:pos + n], streaming=True))
            pos += n
        out["y_stream"] = torch.cat(chunks, -1).numpy()
        out["stream_chunks"] = np.array([5, 7, 11])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hifigan_small.npz"), **out)
    print("hifigan_small: y", tuple(y.shape), "| stream == full:", float((torch.from_numpy(out["y_stream"]) - y[:1].detach()).abs().max()))
    print("\n".join(f"{k} {tuple(v.shape)}" for k, v in list(sd.items())[:12]), "...", len(sd), "keys")


if __name__ == "__main__":
    main()
