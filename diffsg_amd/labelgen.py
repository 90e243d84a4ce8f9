"""Label generator of the MSR problem on the device.

Reference: utils/dataset_generate.py:280-313 `SUM_RATE_GEN(sample_num, M, g_range, W)` ("LRH gradient descent", float64),
used by datasets/sum_rate_gen.py to write the `*c_*w_*samples.csv` training sets.  Same signature and return value
(gs, rates, schemes as numpy float64 arrays); the channel gains are drawn on the host with numpy's global generator exactly
as the reference draws them (or passed in), the 149 descent iterations run in libdiffsg_hip.so (csrc/dsg_labelgen.hpp).
"""
import numpy as np
import torch

from . import _lib


def SUM_RATE_GEN(sample_num=3, M=3, g_range=(0.5, 2.5), W=10.0, gs=None, device=None):
    if gs is None:
        gs = np.random.uniform(g_range[0], g_range[1], size=(sample_num, M))
    gs = np.ascontiguousarray(gs, dtype=np.float64)
    if gs.ndim != 2 or gs.shape[1] != M:
        raise ValueError(f"SUM_RATE_GEN: gs is {gs.shape}, expected (sample_num, {M})")
    if not torch.cuda.is_available():
        raise RuntimeError("SUM_RATE_GEN: no HIP device; libdiffsg_hip has no CPU path")
    dev = torch.device(device if device is not None else "cuda")
    g = torch.from_numpy(gs).to(dev)
    schemes = torch.empty_like(g)
    rates = torch.empty(g.shape[0], device=dev, dtype=torch.float64)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().dsg_sum_rate_gen(_lib.ptr(g), _lib.ptr(schemes), _lib.ptr(rates), g.shape[0], M, float(W), _lib.stream_ptr()))
    return gs, rates.cpu().numpy(), schemes.cpu().numpy()
