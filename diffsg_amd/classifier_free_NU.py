"""Classifier-free-guidance DDPM for NOMA-UAV placement + power (reference: ddpm_opt/classifier_free_NU.py).

Entry points kept: `DDPM`, `nu_data_load`, `train_ddpm_nu`, `custom_decoder`, `rate_calc`, `load_test_nu`; constants
of the reference as defaults (UNet1D(proj 32, dims (32,16,8), n_blocks 2), lr 4e-3, MultiStepLR [80,200], omega 500).
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

DEFAULT_DATASET = "../datasets/3u_18mW_10000samples.csv"


class DDPM(DDPMCore):
    """classifier_free_NU.py:79-127: positional order (T, model, K, P_sum, alphas, device, data_size, ...)."""

    def __init__(self, T, model, K, P_sum, alphas, device, data_size, custom_config=None, uncond_prob=0.1,
                 ema_decay=0.9999, ema_start=1000, ema_update_rate=5, debug=False):
        super().__init__()
        self.K = K
        self.P_sum = P_sum
        self._setup(T, model, alphas, device, data_size, custom_config, uncond_prob, ema_decay, ema_start,
                    ema_update_rate, debug)

    def _decode_recorded(self, i, y):
        """classifier_free_NU.py:174-176 (the reference's branch reads undefined globals; width / height come from
        custom_config here)."""
        from .decode import nu_decode
        cc = self.custom_config or {}
        return nu_decode(y, cc.get("width", 400), cc.get("height", 400), self.P_sum)


def nu_data_load(dataset_path, width, height):
    """classifier_free_NU.py:184-210.  CSV columns: 2K user coords | 2 UAV coords | K powers | 1 rate.  P_sum comes
    from the FILE NAME (`.._<P>mW_..`); coordinates are divided by (width, height), powers by P_sum."""
    src = np.array(pd.read_csv(dataset_path, header=None))
    rows = src.shape[0]
    K = (src.shape[1] - 3) // 3
    P_sum = float(dataset_path.split('_')[-2][:-2])
    X, Y, R = src[:, :2 * K].copy(), src[:, 2 * K:2 + 3 * K].copy(), src[:, -1]
    X[:, 0::2] /= width
    X[:, 1::2] /= height
    Y[:, 0] /= width
    Y[:, 1] /= height
    Y[:, 2:] /= P_sum
    custom_config = {'K': K, 'P_sum': P_sum, 'cdim': 1, 'width': width, 'height': height}
    n_tr, n_te = int(rows * 0.7), int(rows * 0.3)
    return X[:n_tr], Y[:n_tr], X[-n_te:], Y[-n_te:], R[-n_te:], custom_config


def build_model(K, P_sum, device, T=20, custom_config=None):
    """classifier_free_NU.py:228-239."""
    alphas = 1.0 - generate_cosine_schedule(T)
    model = UNet1D(input_dim=2 + K, proj_dim=32, cond_dim=2 * K, dims=(32, 16, 8), is_attn=(False, False, False),
                   middle_attn=False, n_blocks=2)
    return DDPM(T, model, K, P_sum, alphas, device, (1, 2 + K), custom_config, 0.1, 0.9999, 10, 5, False)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device: this build of DiffSG has no CPU path")
    return torch.device("cuda:0")


def train_ddpm_nu(dataset_path=DEFAULT_DATASET, epochs=200, T=20, use_ema=False, warmup_epoch=5, batch_size=512,
                  lr=0.004, milestones=(80, 200), width=400, height=400, log=print):
    """classifier_free_NU.py:213-264."""
    X_train, Y_train, _, _, _, custom_config = nu_data_load(dataset_path, width, height)
    dataset = data.TensorDataset(torch.tensor(X_train, dtype=torch.float32), torch.tensor(Y_train, dtype=torch.float32))
    from .train import FlatAdam, dp_context, make_loader, run_epochs, sync_replicas
    device, rank, world = dp_context()      # one process per GPU when launched under torch.distributed.run; else cuda:0
    loader = make_loader(dataset, batch_size, rank, world)
    if device is None:
        device = _device()                  # raises: no CPU path
    diffusion_model = build_model(custom_config['K'], custom_config['P_sum'], device, T, custom_config)
    diffusion_model.apply(init_weights)
    diffusion_model.to(device)
    sync_replicas(diffusion_model)          # data parallel: rank 0's initial weights everywhere (no-op for one process)
    optimizer = FlatAdam(diffusion_model, lr=lr)  # torch Adam, same update rule, over one flat tensor (one launch)
    scheduler = optim.lr_scheduler.MultiStepLR(optimizer, list(milestones))
    run_epochs(diffusion_model, loader, optimizer, scheduler, epochs, use_ema, warmup_epoch, device, log)
    return diffusion_model


def custom_decoder(Y_pred, width, height, P_sum):
    """classifier_free_NU.py:267-276."""
    from .decode import nu_decode
    return nu_decode(Y_pred, width, height, P_sum)


def rate_calc(Y_pred_decoded, X):
    """classifier_free_NU.py:279-303 (the reference's per-row Python loop, vectorised over rows)."""
    from .decode import nu_rate
    return nu_rate(Y_pred_decoded, X)


@torch.no_grad()
def load_test_nu(ckpt_path, dataset_path=DEFAULT_DATASET, T=20, omega=500, batch_size=512, width=400, height=400,
                 log=print):
    """classifier_free_NU.py:306-361."""
    _, _, X_test, Y_test, _, custom_config = nu_data_load(dataset_path, width, height)
    K, P_sum = custom_config['K'], custom_config['P_sum']
    device = _device()
    diffusion_model = build_model(K, P_sum, device, T, custom_config)
    diffusion_model.load_state_dict(torch.load(ckpt_path, map_location="cpu"))
    diffusion_model.to(device)
    X = torch.tensor(X_test, dtype=torch.float32)
    # the reference's loop of independent `batch_size`-row sample() calls (own noise, own early-step renorm per chunk), run as
    # one set of launches; chunk sizes that are not a multiple of the 32-row tile keep the serial calls
    if batch_size % 32 == 0:
        Y_pred = diffusion_model.sample_chunked_checked(X.to(device), omega, batch_size)
    else:
        Y_pred = torch.cat([diffusion_model.sample_checked(X[i:i + batch_size].to(device), omega) for i in range(0, len(X), batch_size)])
    Xt = X.to(device).clone()
    Xt[:, 0::2] *= width
    Xt[:, 1::2] *= height
    Yd = custom_decoder(Y_pred, width, height, P_sum)
    Yt = torch.tensor(Y_test, dtype=torch.float32, device=device)
    Yt[:, 0] *= width
    Yt[:, 1] *= height
    Yt[:, 2:] *= P_sum
    pred_rate, true_rate = rate_calc(Yd, Xt), rate_calc(Yt, Xt)
    out = {"less_ratio": float(torch.sum(pred_rate) / torch.sum(true_rate)),
           "avg_rate_diff": float(torch.mean(pred_rate - true_rate))}
    log(f"less ratio: {out['less_ratio']}")
    log(f"avg rate diff:\n {out['avg_rate_diff']}")
    return out


def load_test_nu_debug(*args, **kw):
    """classifier_free_NU.py:365-394: record the de-noising trajectory of the test split and write
    `../results/nu_denoise_path.csv` (see trajectory.nu_trajectory_gen_store)."""
    from .trajectory import nu_trajectory_gen_store
    return nu_trajectory_gen_store(*args, **kw)
