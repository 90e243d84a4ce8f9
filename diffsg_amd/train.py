"""Host side of the training step (reference hot loop: classifier_free_MSR.py:217-234, = CO.py:233-250, NU.py:245-262).

`run_epochs` is the reference's epoch loop; the per-batch arithmetic (q_sample, denoiser forward and backward, MSE)
is DDPM.forward -> libdiffsg_hip.so.  Adam and MultiStepLR stay PyTorch objects (not part of the path).
With torch.distributed initialised, gradients of `model.*` are averaged over ranks with ONE all-reduce per step over a
flat float32 bucket (RCCL over xGMI on MI355X; gloo in the CPU tests); `ema.module.*` never receives gradients and is
not communicated.
"""
from __future__ import annotations



def run_epochs(diffusion_model, loader, optimizer, scheduler, epochs, use_ema, warmup_epoch, device, log=print):
    ema_step_cnt = 1
    for epoch in range(epochs):
        epoch_loss, epoch_rows = 0.0, 0
        for x, y_true in loader:
            x = x.to(device)
            y_true = y_true.to(device)
            loss = diffusion_model(y_true, x)
            loss.backward()
            diffusion_model.allreduce_grads()   # no-op unless torch.distributed is initialised with world_size > 1
            optimizer.step()
            optimizer.zero_grad()
            if (use_ema and epoch > warmup_epoch and ema_step_cnt > diffusion_model.ema_start
                    and ema_step_cnt % diffusion_model.ema_update_rate == 0):
                diffusion_model.ema.update_parameters(diffusion_model.model)
            epoch_loss += loss.item()
            epoch_rows += x.shape[0]
            ema_step_cnt += 1
        # the reference prints (sum of batch-mean losses) / (row count), MSR.py:233 -- reproduced as is
        log(f"Epoch: {epoch}, Loss: {epoch_loss / epoch_rows}")
        scheduler.step()
    return diffusion_model
