"""Schedule and weight initialisation (reference: ddpm_opt/diffusion.py:17-35 and :82-84).

Only the two functions the classifier-free scripts import are provided; the reference's `DiffusionOpt` class is dead
code (naming-mismatch note).  Host-side float64 numpy, run once per model.
"""
import numpy as np
import torch.nn as nn


def generate_cosine_schedule(T, s=0.008):
    """betas[T] (float64).  abar_t = f(t)/f(0), f(t) = cos^2(((t/T + s)/(1 + s)) * pi/2); beta_t = min(1 - abar_t/abar_{t-1}, 0.84)."""
    t = np.arange(T + 1, dtype=np.float64)
    # evaluated element by element in the reference; numpy's vector cos/divide give the same float64 values
    f = np.cos((t / T + s) / (1 + s) * np.pi / 2) ** 2
    abar = f / f[0]
    return np.minimum(1 - abar[1:] / abar[:-1], 0.84)


def init_weights(m):
    """`model.apply(init_weights)`: every nn.Linear weight ~ N(0, 0.01^2); biases and LayerNorms keep torch defaults."""
    if type(m) is nn.Linear:
        nn.init.normal_(m.weight, std=0.01)
