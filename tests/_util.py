"""Helpers shared by the test modules (tests/ is on sys.path in pytest's prepend mode)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLD):
    if p not in sys.path:
        sys.path.insert(0, p)


def synth_params(cfg_name, seed, flavour="trained"):
    """{key: torch.float32 tensor} of deterministic synthetic weights for a named config."""
    from weights import CONFIGS, synth_weights
    from oracle import ddpm_oracle as O
    cfg = CONFIGS[cfg_name]
    plan = O.unet_plan(cfg["input_dim"], cfg["proj_dim"], cfg["cond_dim"], cfg["dims"], cfg["n_blocks"])
    w = synth_weights(O.state_shapes(plan), seed, flavour)
    return plan, {k: torch.from_numpy(v) for k, v in w.items()}
