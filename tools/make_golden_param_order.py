#!/usr/bin/env python3
"""Fixture: the ORDER of the reference model's parameters and state-dict keys, by instantiating the reference modules here.

    python tools/make_golden_param_order.py        # writes tests/golden/ref_param_order.npz   (names only, no weights)

A Lightning checkpoint stores optimizer state by parameter INDEX in the order of ``module.parameters()``
(base_lightning_module.py:47-55: one AdamW over ``self.generator.parameters()``, one over ``self.discriminator.parameters()``);
importing it needs that order.  The names are listed for the generator (BASELINE-shaped key set; the order does not depend on
the widths) and for VocosDiscriminator.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import tools.make_golden as MG  # noqa: E402  (installs the stubs, imports the reference modules)
from oracle import schema as S  # noqa: E402

gen = MG.build_generator(S.SMALL)
disc, _ = MG.build_disc(0)
out = {"generator_params": np.array([k for k, _ in gen.named_parameters()]),
       "generator_state_keys": np.array(list(gen.state_dict().keys())),
       "discriminator_params": np.array([k for k, _ in disc.named_parameters()]),
       "discriminator_state_keys": np.array(list(disc.state_dict().keys()))}
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_param_order.npz"), **out)
print({k: len(v) for k, v in out.items()})
# the assumption the importer relies on: parameters() order == state_dict() order restricted to the parameters
for nm, mod in (("generator", gen), ("discriminator", disc)):
    params = [k for k, _ in mod.named_parameters()]
    sd = [k for k in mod.state_dict().keys() if k in set(params)]
    assert params == sd, nm
print("parameters() order == state_dict() order on the parameter keys: OK")
