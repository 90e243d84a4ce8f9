"""Exponential moving average of the denoiser weights (reference: ddpm_opt/ema.py:3-14).

The reference subclasses torch.optim.swa_utils.AveragedModel(model, device, avg_fn, use_buffers=True); what the hot
path needs from it is: a deep copy under `.module`, an int64 buffer `n_averaged`, state-dict keys `n_averaged`,
`module.*`, and `update_parameters(model)` = copy on the first call, `avg = decay*avg + (1-decay)*p` afterwards.
The update runs as one HIP axpby per tensor (dsg_ema_update) when the tensors live on the GPU.
"""
import copy

import torch
import torch.nn as nn

from . import _lib


class ExponentialMovingAverage(nn.Module):
    def __init__(self, model, decay, device="cpu"):
        super().__init__()
        self.register_buffer("n_averaged", torch.tensor(0, dtype=torch.long))
        self.module = copy.deepcopy(model)
        if device is not None:
            self.module = self.module.to(device)
            self.n_averaged = self.n_averaged.to(device)
        self.decay = float(decay)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @torch.no_grad()
    def update_parameters(self, model):
        avg = list(self.module.parameters()) + list(self.module.buffers())
        cur = list(model.parameters()) + list(model.buffers())
        first = int(self.n_averaged) == 0
        for a, p in zip(avg, cur):
            p = p.detach()
            if first:
                a.copy_(p)
            elif a.is_cuda:
                p = p.to(a.device).contiguous()
                with torch.cuda.device(a.device):
                    _lib.check(_lib.lib().dsg_ema_update(_lib.ptr(a), _lib.ptr(p), self.decay, 1.0 - self.decay,
                                                         a.numel(), _lib.stream_ptr()))
            else:
                raise RuntimeError("ExponentialMovingAverage.update_parameters: averaged weights are not on a HIP "
                                   "device; there is no CPU path")
        if hasattr(self.module, "mark_weights_changed"):
            self.module.mark_weights_changed()     # dsg_ema_update writes through raw pointers: tensor versions do not move
        self.n_averaged += 1
