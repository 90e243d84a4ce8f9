"""Deterministic synthetic denoiser weights shared by make_goldens.py and the tests.

Weights are never stored in the fixtures (1.5 M floats per config); they are
regenerated from ``numpy.random.RandomState(seed)`` (whose stream numpy
guarantees to keep stable) for a given ``{key: shape}`` table.  Two flavours:

* ``"trained"``  - activations stay O(1) through the net: Linear.weight ~
  N(0, 1/fan_in), biases ~ 0.1 N, LayerNorm gamma ~ 1 + 0.1 N, beta ~ 0.1 N.
* ``"init"``     - the state right after the reference's ``init_weights``
  (diffusion.py:82-84): Linear.weight ~ N(0, 0.01); torch-default-like biases
  U(-1/sqrt(fan_in), 1/sqrt(fan_in)); LayerNorm (1, 0).
"""
import numpy as np

CONFIGS = {
    # name: UNet1D kwargs
    "msr3":  dict(input_dim=3,  proj_dim=128, cond_dim=3,  dims=(64, 32, 16, 8), n_blocks=2),
    "msr80": dict(input_dim=80, proj_dim=128, cond_dim=80, dims=(64, 32, 16, 8), n_blocks=2),
    "co3":   dict(input_dim=3,  proj_dim=64,  cond_dim=9,  dims=(64, 32, 16, 8), n_blocks=3),
    "nu3":   dict(input_dim=5,  proj_dim=32,  cond_dim=6,  dims=(32, 16, 8),     n_blocks=2),
    # the reference's debug-size net (classifier_free_MSR.py:315-316)
    "tiny":  dict(input_dim=3,  proj_dim=16,  cond_dim=3,  dims=(16, 8, 4),      n_blocks=2),
}


def synth_weights(shapes, seed, flavour="trained"):
    """{key: float32 ndarray} for an ordered ``{key: shape}`` table."""
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in shapes.items():
        is_norm = ".norm" in key or key.startswith("norm.")
        n = rs.standard_normal(shape)
        if flavour == "trained":
            if is_norm:
                v = 1.0 + 0.1 * n if key.endswith(".weight") else 0.1 * n
            elif key.endswith(".weight"):
                v = n / np.sqrt(shape[1])
            else:
                v = 0.1 * n
        elif flavour == "init":
            if is_norm:
                v = np.ones(shape) if key.endswith(".weight") else np.zeros(shape)
            elif key.endswith(".weight"):
                v = 0.01 * n
            else:
                # bias: fan_in is unknown from the bias shape alone; take it from the weight drawn just before
                fan_in = out[key[:-len("bias")] + "weight"].shape[1]
                v = (rs.uniform(-1.0, 1.0, shape)) / np.sqrt(fan_in)
        else:
            raise ValueError(flavour)
        out[key] = np.ascontiguousarray(v, dtype=np.float32)
    return out
