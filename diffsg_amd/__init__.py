"""diffsg_amd -- MI355X-native (gfx950) implementation of DiffSG's classifier-free DDPM hot path.

Host-side mirror of the reference's Python object API (ddpm_opt/*.py) over the C ABI of libdiffsg_hip.so
(include/diffsg.h).  See DESIGN.md.
"""
from .UNetCF import UNet1D  # noqa: F401
from .diffusion import generate_cosine_schedule, init_weights  # noqa: F401
from .ema import ExponentialMovingAverage  # noqa: F401

__all__ = ["UNet1D", "generate_cosine_schedule", "init_weights", "ExponentialMovingAverage"]
