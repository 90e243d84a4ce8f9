"""Classifier-free-guidance DDPM for Computation Offloading (reference: ddpm_opt/classifier_free_CO.py).

Entry points kept: `DDPM`, `co_data_load`, `train_ddpm_co`, `cost_calc`, `customized_real_decoder`, `load_test_co`, and
the self-check harness `validation_data_gen` / `validate_ddpm_co` / `test_ddpm`;
constants of the reference as defaults (UNet1D(proj 64, dims (64,32,16,8), n_blocks 3), lr 5e-3,
MultiStepLR [15,80,150], omega 500).
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

DEFAULT_DATASET = "../datasets/3nodes_50000samples_new.csv"


class DDPM(DDPMCore):
    """classifier_free_CO.py:55-98: positional order (T, model, node_num, alphas, device, data_size, ...)."""

    def __init__(self, T, model, node_num, alphas, device, data_size, custom_config=None, uncond_prob=0.1,
                 ema_decay=0.9999, ema_start=1000, ema_update_rate=5, debug=False):
        super().__init__()
        self.node_num = node_num
        self._setup(T, model, alphas, device, data_size, custom_config, uncond_prob, ema_decay, ema_start,
                    ema_update_rate, debug)

    def _decode_recorded(self, i, y):
        """classifier_free_CO.py:94-96."""
        from .decode import co_decode
        return co_decode(y)


def data_preprocess_co(X):
    """utils/dataset.py:26-51: (6 per-node + 7 common raw features) -> 3 cost features per node
    (local cost, offload transition cost, ideal offload execution cost)."""
    node_num = (X.shape[1] - 7) // 6
    F_t, kappa, Pt, PI, theta, Bw, N0 = (X[:, -7 + i] for i in range(7))
    col = lambda i, j: X[:, 6 * i + j]
    interference = sum(Pt * col(i, 3) ** 2 for i in range(node_num))
    out = np.zeros((X.shape[0], node_num * 3))
    for i in range(node_num):
        d, c, f, hgain, a = col(i, 0), col(i, 1), col(i, 2), col(i, 3), col(i, 4)
        r_u = Bw * np.log2(1.0 + Pt * hgain ** 2 / (N0 + interference))
        out[:, 3 * i] = a * c / f + (1.0 - a) * kappa * f ** 2 * c
        out[:, 3 * i + 1] = a * d / r_u + (1.0 - a) * Pt * d / r_u
        out[:, 3 * i + 2] = a * c / F_t + (1.0 - a) * PI * c / F_t
    return out


_COMMON = np.array([[2.5e9, 1e-28, 0.3, 0.1, 1.0, 10e5, 7.96159e-13]], dtype=float)  # F_t kappa Pt PI theta B N0


def co_data_load(dataset_path):
    """classifier_free_CO.py:158-200.  Rows with any simplified feature >= 10 are dropped; the 70/30 split sizes
    come from the PRE-filter row count (:198-199)."""
    src = np.array(pd.read_csv(dataset_path, header=None))
    rows = src.shape[0]
    node_num = (src.shape[1] - 1) // 7
    X, Y = src[:, :6 * node_num], src[:, -node_num:]
    X = data_preprocess_co(np.concatenate((X, np.tile(_COMMON, (rows, 1))), axis=1))
    keep = np.all(X < 10.0, axis=1)
    X, Y = X[keep], Y[keep]
    lo, hi = np.min(X), np.max(X)
    X = (X - lo) / (hi - lo)
    custom_config = {'sfn': 3, 'cfn': 0, 'cdim': 1, 'scaler_min': lo, 'scaler_max': hi}
    n_tr, n_te = int(rows * 0.7), int(rows * 0.3)
    return X[:n_tr], Y[:n_tr], X[-n_te:], Y[-n_te:], custom_config


def build_model(node_num, cond_dim, device, T=20, custom_config=None):
    """classifier_free_CO.py:216-227."""
    alphas = 1.0 - generate_cosine_schedule(T)
    model = UNet1D(input_dim=node_num, proj_dim=64, cond_dim=cond_dim, dims=(64, 32, 16, 8),
                   is_attn=(False, False, False, False), middle_attn=False, n_blocks=3)
    return DDPM(T, model, node_num, alphas, device, (1, node_num), custom_config, 0.1, 0.9999, 10, 5, False)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device: this build of DiffSG has no CPU path")
    return torch.device("cuda:0")


def train_ddpm_co(dataset_path=DEFAULT_DATASET, epochs=200, T=20, use_ema=False, warmup_epoch=5, batch_size=512,
                  lr=0.005, milestones=(15, 80, 150), log=print):
    """classifier_free_CO.py:203-252."""
    X_train, Y_train, _, _, custom_config = co_data_load(dataset_path)
    dataset = data.TensorDataset(torch.tensor(X_train, dtype=torch.float32), torch.tensor(Y_train, dtype=torch.float32))
    from .train import FlatAdam, dp_context, make_loader, run_epochs, sync_replicas
    device, rank, world = dp_context()      # one process per GPU when launched under torch.distributed.run; else cuda:0
    loader = make_loader(dataset, batch_size, rank, world)
    node_num = Y_train.shape[1]
    if device is None:
        device = _device()                  # raises: no CPU path
    diffusion_model = build_model(node_num, custom_config['sfn'] * node_num, device, T, custom_config)
    diffusion_model.apply(init_weights)
    diffusion_model.to(device)
    sync_replicas(diffusion_model)          # data parallel: rank 0's initial weights everywhere (no-op for one process)
    optimizer = FlatAdam(diffusion_model, lr=lr)  # torch Adam, same update rule, over one flat tensor (one launch)
    scheduler = optim.lr_scheduler.MultiStepLR(optimizer, list(milestones))
    run_epochs(diffusion_model, loader, optimizer, scheduler, epochs, use_ema, warmup_epoch, device, log)
    return diffusion_model


def cost_calc(X, Y):
    """classifier_free_CO.py:255-278."""
    from .decode import co_cost
    return co_cost(X, Y)


def customized_real_decoder(Y_pred):
    """classifier_free_CO.py:281-290."""
    from .decode import co_decode
    return co_decode(Y_pred)


@torch.no_grad()
def load_test_co(ckpt_path, dataset_path=DEFAULT_DATASET, T=20, omega=500.0, batch_size=512, log=print):
    """classifier_free_CO.py:293-356."""
    X_train, Y_train, X_test, Y_test, custom_config = co_data_load(dataset_path)
    node_num = Y_train.shape[1]
    device = _device()
    diffusion_model = build_model(node_num, custom_config['sfn'] * node_num, device, T, custom_config)
    diffusion_model.load_state_dict(torch.load(ckpt_path, map_location="cpu"))
    diffusion_model.to(device)
    X = torch.tensor(X_test, dtype=torch.float32)
    # the reference's loop of independent `batch_size`-row sample() calls (own noise, own early-step renorm per chunk), run as
    # one set of launches; chunk sizes that are not a multiple of the 32-row tile keep the serial calls
    if batch_size % 32 == 0:
        Y_pred = diffusion_model.sample_chunked_checked(X.to(device), omega, batch_size)
    else:
        Y_pred = torch.cat([diffusion_model.sample_checked(X[i:i + batch_size].to(device), omega) for i in range(0, len(X), batch_size)])
    Xt = X.to(device) * (custom_config['scaler_max'] - custom_config['scaler_min']) + custom_config['scaler_min']
    Yt = torch.tensor(Y_test, dtype=torch.float32, device=device)
    Yd = customized_real_decoder(Y_pred)
    pred_cost, true_cost = cost_calc(Xt, Yd), cost_calc(Xt, Yt)
    weights = 2 ** torch.arange(node_num - 1, -1, -1, device=device)
    pred_cls = ((Yd > 0.1).long() * weights).sum(dim=1)
    true_cls = ((Yt > 0.1).long() * weights).sum(dim=1)
    terrible = ((pred_cost / true_cost > 1.2) & (pred_cost > 10.0)).sum()
    out = {"exceeded_ratio": float(torch.sum(pred_cost) / torch.sum(true_cost)),
           "avg_cost_diff": float(torch.mean(pred_cost - true_cost)),
           "terrible": int(terrible), "accuracy": int((pred_cls == true_cls).sum()), "n": int(X.shape[0])}
    log(f"exceeded ratio: {out['exceeded_ratio']}")
    log(f"avg cost diff:\n {out['avg_cost_diff']}")
    log(f"terrible samples num: {out['terrible']}/{out['n']}.")
    log(f"accuracy: {out['accuracy']}/{out['n']}")
    return out


@torch.no_grad()
def load_test_co_debug(ckpt_path, dataset_path=DEFAULT_DATASET, T=400, omega=150.0, batch_size=512, want2look=tuple(range(10)), log=print):
    """classifier_free_CO.py:358-413: the first test batch with the de-noising path recorded; prints, for the wanted rows, the
    condition, the label and every step's (decoded y_t, guided eps).  Returns (Y_pred, y_record, eps_record) of that batch."""
    X_train, Y_train, X_test, Y_test, custom_config = co_data_load(dataset_path)
    node_num = Y_train.shape[1]
    device = _device()
    diffusion_model = build_model(node_num, custom_config['sfn'] * node_num, device, T, custom_config)
    diffusion_model.load_state_dict(torch.load(ckpt_path, map_location="cpu"))
    diffusion_model.to(device)
    x = torch.tensor(X_test[:batch_size], dtype=torch.float32, device=device)
    diffusion_model.record_denoise_path = True
    Y_pred = diffusion_model.sample_checked(x, omega)
    diffusion_model.record_denoise_path = False
    ys = diffusion_model.y_i_record.reshape(x.shape[0], T, -1)
    es = diffusion_model.eps_i_record.reshape(x.shape[0], T, -1)
    for i in want2look:
        if i >= x.shape[0]:
            break
        log("%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%")
        log(x[i], Y_test[i])
        for j in range(T):
            log(ys[i, j, :], es[i, j, :])
    return Y_pred, ys, es


# ------------------------------------------------------------------ self-check harness (classifier_free_CO.py:416-558)
def validation_data_gen():
    """classifier_free_CO.py:416-449: 3 x 1000 rows whose label is the one-hot index of the node block that got +1; same
    numpy draws in the same order (random((1000, 3)), then permutation(3000)), 70/30 split."""
    X_base = np.random.random((1000, 3))
    blocks = []
    for k in range(3):
        X = np.concatenate([X_base + 1 if i == k else X_base for i in range(3)], axis=1)
        Y = np.zeros((1000, 3))
        Y[:, k] = 1
        blocks.append(np.concatenate((Y, X), axis=1))
    src = np.concatenate(blocks, axis=0)
    src = src[np.random.permutation(src.shape[0])]
    X, Y = src[:, 3:], src[:, :3]
    n_tr, n_te = int(src.shape[0] * 0.7), int(src.shape[0] * 0.3)
    return X[:n_tr], Y[:n_tr], X[-n_te:], Y[-n_te:], {'sfn': 3, 'cfn': 0}


def _validation_model(node_num, cond_dim, device, T, custom_config, uncond_prob):
    """UNet1D(proj 64, dims (32,16,8), n_blocks 2) of classifier_free_CO.py:468-476 / :521-528."""
    alphas = 1.0 - generate_cosine_schedule(T)
    model = UNet1D(input_dim=node_num, proj_dim=64, cond_dim=cond_dim, dims=(32, 16, 8), is_attn=(False, False, False),
                   middle_attn=False, n_blocks=2)
    return DDPM(T, model, node_num, alphas, device, (1, 3), custom_config, uncond_prob, 0.9999, 10, 5, False)


def validate_ddpm_co(epochs=500, T=500, use_ema=False, warmup_epoch=5, batch_size=512, lr=0.005,
                     milestones=(30, 150, 350), data_split=None, log=print):
    """classifier_free_CO.py:451-502: train on the validation set (uncond_prob 0.0, MultiStepLR [30,150,350])."""
    X_train, Y_train, _, _, custom_config = data_split if data_split is not None else validation_data_gen()
    dataset = data.TensorDataset(torch.tensor(X_train, dtype=torch.float32), torch.tensor(Y_train, dtype=torch.float32))
    from .train import FlatAdam, dp_context, make_loader, run_epochs, sync_replicas
    device, rank, world = dp_context()      # one process per GPU when launched under torch.distributed.run; else cuda:0
    loader = make_loader(dataset, batch_size, rank, world)
    node_num = Y_train.shape[1]
    if device is None:
        device = _device()
    diffusion_model = _validation_model(node_num, custom_config['sfn'] * node_num, device, T, custom_config, 0.0)
    diffusion_model.apply(init_weights)
    diffusion_model.to(device)
    sync_replicas(diffusion_model)          # every replica starts from rank 0's draw (as train_ddpm_co)
    optimizer = FlatAdam(diffusion_model, lr=lr)
    scheduler = optim.lr_scheduler.MultiStepLR(optimizer, list(milestones))
    run_epochs(diffusion_model, loader, optimizer, scheduler, epochs, use_ema, warmup_epoch, device, log)
    return diffusion_model


@torch.no_grad()
def test_ddpm(ckpt_path=None, T=500, omega=30.0, batch_size=512, diffusion_model=None, data_split=None, log=print):
    """classifier_free_CO.py:504-558: sample the validation test split, softmax, threshold 0.1, compare the decision
    pattern (as a binary number over the nodes) with the label's; returns and prints `accuracy: k/n`."""
    _, Y_train, X_test, Y_test, custom_config = data_split if data_split is not None else validation_data_gen()
    node_num = Y_train.shape[1]
    device = _device()
    if diffusion_model is None:
        diffusion_model = _validation_model(node_num, custom_config['sfn'] * node_num, device, T, custom_config, 0.1)
        diffusion_model.load_state_dict(torch.load(ckpt_path, map_location="cpu"))
    diffusion_model.to(device)
    X = torch.tensor(X_test, dtype=torch.float32)
    # the reference's loop of independent `batch_size`-row sample() calls (own noise, own early-step renorm per chunk), run as
    # one set of launches; chunk sizes that are not a multiple of the 32-row tile keep the serial calls
    if batch_size % 32 == 0:
        Y_pred = diffusion_model.sample_chunked_checked(X.to(device), omega, batch_size)
    else:
        Y_pred = torch.cat([diffusion_model.sample_checked(X[i:i + batch_size].to(device), omega) for i in range(0, len(X), batch_size)])
    from .decode import row_softmax
    Y_pred = row_softmax(Y_pred)
    Yt = torch.tensor(Y_test, dtype=torch.float32, device=device)
    weights = 2 ** torch.arange(node_num - 1, -1, -1, device=device)
    pred_cls = ((Y_pred > 0.1).long() * weights).sum(dim=1)
    true_cls = ((Yt > 0.1).long() * weights).sum(dim=1)
    hits = int((pred_cls == true_cls).sum())
    log(f"accuracy: {hits}/{X.shape[0]}")
    return {"accuracy": hits, "n": int(X.shape[0]), "Y_pred": Y_pred}
