"""Classifier-free-guidance DDPM for Maximum Sum Rate power allocation (reference: ddpm_opt/classifier_free_MSR.py).

Same entry points as the reference script -- `DDPM`, `msr_data_load`, `train_ddpm_msr`, `custom_decoder`,
`load_test_msr` -- with the reference's constants as defaults (T=20, batch 512, Adam lr 5e-3, MultiStepLR [100,150],
omega=500, UNet1D(proj 128, dims (64,32,16,8), n_blocks 2)).  The denoiser, the reverse loop and the training step
run in libdiffsg_hip.so; this file is host orchestration.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch
import torch.optim as optim
import torch.utils.data as data

from .ddpm import DDPMCore
from .diffusion import generate_cosine_schedule, init_weights
from .UNetCF import UNet1D

DEFAULT_DATASET = "../datasets/3c_10w_10000samples.csv"


class DDPM(DDPMCore):
    """classifier_free_MSR.py:50-98: positional order (T, model, M, W, alphas, device, data_size, ...)."""

    def __init__(self, T, model, M, W, alphas, device, data_size, custom_config=None, uncond_prob=0.1,
                 ema_decay=0.9999, ema_start=1000, ema_update_rate=5, debug=False):
        super().__init__()
        self.M = M
        self.W = W
        self._setup(T, model, alphas, device, data_size, custom_config, uncond_prob, ema_decay, ema_start,
                    ema_update_rate, debug)

    def _decode_recorded(self, i, y):
        """classifier_free_MSR.py:145-151: row softmax for the first three recorded states, the MSR decoder afterwards."""
        from .decode import msr_decode
        from .decode import row_softmax
        return row_softmax(y) if i <= 2 else msr_decode(y)


def msr_data_load(dataset_path):
    """classifier_free_MSR.py:159-184.  CSV columns: M gains | 1 rate | M powers.  W comes from the FILE NAME
    (`.._<W>w_..`), X is min-max scaled with the GLOBAL min/max, split = first 70 % / last 30 % of the rows."""
    src = np.array(pd.read_csv(dataset_path, header=None))
    rows = src.shape[0]
    M = (src.shape[1] - 1) // 2
    W = float(dataset_path.split('_')[-2][:-1])
    X, Y = src[:, :M], src[:, -M:]
    lo, hi = np.min(X), np.max(X)
    X = (X - lo) / (hi - lo)
    custom_config = {'M': M, 'W': W, 'sfn': 1, 'cfn': 0, 'cdim': 1, 'scaler_min': lo, 'scaler_max': hi}
    n_tr, n_te = int(rows * 0.7), int(rows * 0.3)
    return X[:n_tr], Y[:n_tr], X[-n_te:], Y[-n_te:], custom_config


def build_model(M, cond_dim, device, T=20, custom_config=None, W=10.0):
    """UNet1D + DDPM exactly as classifier_free_MSR.py:200-211 builds them (hyper-parameters of :202-203)."""
    alphas = 1.0 - generate_cosine_schedule(T)
    model = UNet1D(input_dim=M, proj_dim=128, cond_dim=cond_dim, dims=(64, 32, 16, 8),
                   is_attn=(False, False, False, False), middle_attn=False, n_blocks=2)
    return DDPM(T, model, M, W, alphas, device, (1, M), custom_config, 0.1, 0.9999, 10, 5, False)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device: this build of DiffSG has no CPU path")
    return torch.device("cuda:0")


def train_ddpm_msr(dataset_path=DEFAULT_DATASET, epochs=200, T=20, use_ema=False, warmup_epoch=5, batch_size=512,
                   lr=0.005, milestones=(100, 150), log=print):
    """classifier_free_MSR.py:187-236 (hot loop :217-234)."""
    X_train, Y_train, _, _, custom_config = msr_data_load(dataset_path)
    dataset = data.TensorDataset(torch.tensor(X_train, dtype=torch.float32), torch.tensor(Y_train, dtype=torch.float32))
    from .train import FlatAdam, dp_context, make_loader, run_epochs, sync_replicas
    device, rank, world = dp_context()      # one process per GPU when launched under torch.distributed.run; else cuda:0
    loader = make_loader(dataset, batch_size, rank, world)
    M, W = custom_config['M'], custom_config['W']
    if device is None:
        device = _device()                  # raises: no CPU path
    diffusion_model = build_model(M, custom_config['sfn'] * M, device, T, custom_config, W)
    diffusion_model.apply(init_weights)
    diffusion_model.to(device)
    sync_replicas(diffusion_model)          # data parallel: rank 0's initial weights everywhere (no-op for one process)
    optimizer = FlatAdam(diffusion_model, lr=lr)  # torch Adam, same update rule, over one flat tensor (one launch)
    scheduler = optim.lr_scheduler.MultiStepLR(optimizer, list(milestones))
    run_epochs(diffusion_model, loader, optimizer, scheduler, epochs, use_ema, warmup_epoch, device, log)
    return diffusion_model


def custom_decoder(Y_pred):
    """classifier_free_MSR.py:239-245: global min-max over the whole tensor, then a row softmax."""
    from .decode import msr_decode
    return msr_decode(Y_pred)


@torch.no_grad()
def load_test_msr(ckpt_path, dataset_path=DEFAULT_DATASET, T=20, omega=500, batch_size=512, log=print):
    """classifier_free_MSR.py:248-298; returns the printed metrics as a dict as well."""
    _, _, X_test, Y_test, custom_config = msr_data_load(dataset_path)
    M, W = custom_config['M'], custom_config['W']
    device = _device()
    diffusion_model = build_model(M, custom_config['sfn'] * M, device, T, custom_config, W)
    diffusion_model.load_state_dict(torch.load(ckpt_path, map_location="cpu"))
    diffusion_model.to(device)
    X = torch.tensor(X_test, dtype=torch.float32)
    # each 512-row chunk is its own sample() call, as in the reference (:273-279): the early-step renorm is per call
    # the reference's loop of independent `batch_size`-row sample() calls (own noise, own early-step renorm per chunk), run as
    # one set of launches; chunk sizes that are not a multiple of the 32-row tile keep the serial calls
    if batch_size % 32 == 0:
        Y_pred = diffusion_model.sample_chunked_checked(X.to(device), omega, batch_size)
    else:
        Y_pred = torch.cat([diffusion_model.sample_checked(X[i:i + batch_size].to(device), omega) for i in range(0, len(X), batch_size)])
    Xt = X.to(device) * (custom_config['scaler_max'] - custom_config['scaler_min']) + custom_config['scaler_min']
    Yt = torch.tensor(Y_test, dtype=torch.float32, device=device)
    from .decode import msr_rate
    Yd = W * custom_decoder(Y_pred)
    pred_rate, true_rate = msr_rate(Yd, Xt), msr_rate(Yt, Xt)
    out = {"less_ratio": float(torch.sum(pred_rate) / torch.sum(true_rate)),
           "avg_rate_diff": float(torch.mean(pred_rate - true_rate))}
    log(f"less ratio: {out['less_ratio']}")
    log(f"avg rate diff:\n {out['avg_rate_diff']}")
    return out


@torch.no_grad()
def load_test_msr_debug(ckpt_path, dataset_path=DEFAULT_DATASET, T=20, omega=150, want2look=tuple(range(10)), log=print):
    """classifier_free_MSR.py:301-344: sample a handful of test rows with the de-noising path recorded and print, per row, the
    condition, the prediction and every step's (decoded y_t, guided eps).  The reference indexes its records `[step, row]`; here
    they are `(rows, T*D)` (MSR.py:153-154), so the loop reshapes.  Returns (Y_pred, y_record, eps_record)."""
    _, _, X_test, _, custom_config = msr_data_load(dataset_path)
    M, W = custom_config['M'], custom_config['W']
    device = _device()
    diffusion_model = build_model(M, custom_config['sfn'] * M, device, T, custom_config, W)
    diffusion_model.load_state_dict(torch.load(ckpt_path, map_location="cpu"))
    diffusion_model.to(device)
    X = torch.tensor(X_test, dtype=torch.float32, device=device)[list(want2look)]
    diffusion_model.record_denoise_path = True
    Y_pred = diffusion_model.sample_checked(X, omega)
    diffusion_model.record_denoise_path = False
    ys = diffusion_model.y_i_record.reshape(X.shape[0], T, -1)
    es = diffusion_model.eps_i_record.reshape(X.shape[0], T, -1)
    for i in range(X.shape[0]):
        log("%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%")
        log(X[i], Y_pred[i])
        for j in range(T):
            log(ys[i, j, :], es[i, j, :])
    return Y_pred, ys, es

