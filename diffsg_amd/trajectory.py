"""De-noising trajectory files (reference: datasets/sum_rate_trajectory_gen.py:25-51, datasets/co_trajectory_gen.py:20-59,
ddpm_opt/classifier_free_NU.py:365-394) and the sum-rate dataset file (datasets/sum_rate_gen.py:9-14).

The reference records the trajectory with two device-to-host copies per reverse step (MSR.py:139-141); here every step
lands in a device-side ring inside the per-step graph (dsg_sample_rec) and comes back with ONE copy per `sample()` call.
File format kept: one row per test sample, `T * D` columns = the decoded y after each reverse step (step T-1 first),
no header, no index.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch


def _load(diffusion_model, ckpt_path, device):
    if ckpt_path is not None:
        diffusion_model.load_state_dict(torch.load(ckpt_path, map_location="cpu"))
    diffusion_model.to(device)
    diffusion_model.record_denoise_path = True
    return diffusion_model


@torch.no_grad()
def record_trajectories(diffusion_model, X_test, omega, batch_size=None):
    """(rows, T*D) float array of decoded trajectories.  batch_size=None: one `sample()` call over all rows (the MSR / NU
    scripts); an int: chunks of that many rows written into place (co_trajectory_gen.py:48-56)."""
    dev = next(diffusion_model.model.parameters()).device
    X = torch.as_tensor(np.asarray(X_test), dtype=torch.float32)
    if batch_size is None:
        diffusion_model.sample_checked(X.to(dev), omega)
        return np.asarray(diffusion_model.y_i_record)
    D = diffusion_model.model.cfg["input_dim"]
    out = np.zeros((X.shape[0], D * diffusion_model.T), dtype=float)
    for i in range(0, X.shape[0], batch_size):
        diffusion_model.sample_checked(X[i:i + batch_size].to(dev), omega)
        out[i:i + batch_size, :] = diffusion_model.y_i_record
    return out


def _store(traj, out_csv, log):
    pd.DataFrame(traj).to_csv(out_csv, header=None, index=False)
    log(f"Trajectory generating finished, {traj.shape[0]} samples stored.")
    return traj


@torch.no_grad()
def msr_trajectory_gen_store(ckpt_path="../ckpts/ddpm_msr_3c.pt", dataset_path="../datasets/3c_10w_10000samples.csv",
                             out_csv="../results/msr_denoise_path.csv", T=20, omega=500, diffusion_model=None, log=print):
    """sum_rate_trajectory_gen.py:25-51 (UNet1D(proj 128, dims (64,32,16,8), n_blocks 2), omega 500, whole test split)."""
    from . import classifier_free_MSR as M
    _, _, X_test, _, cfg = M.msr_data_load(dataset_path)
    dev = M._device()
    if diffusion_model is None:
        diffusion_model = M.build_model(cfg['M'], cfg['sfn'] * cfg['M'], dev, T, cfg, cfg['W'])
    return _store(record_trajectories(_load(diffusion_model, ckpt_path, dev), X_test, omega), out_csv, log)


@torch.no_grad()
def co_trajectory_gen_store(ckpt_path="../ckpts/ddpm_co.pt", dataset_path="../datasets/3nodes_50000samples_new.csv",
                            out_csv="../results/co_denoise_path.csv", T=20, omega=500, batch_size=512, diffusion_model=None,
                            log=print):
    """co_trajectory_gen.py:20-59 (512-row chunks of the test split)."""
    from . import classifier_free_CO as C
    _, Y_train, X_test, _, cfg = C.co_data_load(dataset_path)
    dev = C._device()
    node_num = Y_train.shape[1]
    if diffusion_model is None:
        diffusion_model = C.build_model(node_num, cfg['sfn'] * node_num, dev, T, cfg)
    return _store(record_trajectories(_load(diffusion_model, ckpt_path, dev), X_test, omega, batch_size), out_csv, log)


@torch.no_grad()
def nu_trajectory_gen_store(ckpt_path="../ckpts/ddpm_nu_3u.pt", dataset_path="../datasets/3u_18mW_10000samples.csv",
                            out_csv="../results/nu_denoise_path.csv", T=20, omega=500, width=400, height=400,
                            diffusion_model=None, log=print):
    """classifier_free_NU.py:365-394 (`load_test_nu_debug`)."""
    from . import classifier_free_NU as N
    _, _, X_test, _, _, cfg = N.nu_data_load(dataset_path, width, height)
    dev = N._device()
    if diffusion_model is None:
        diffusion_model = N.build_model(cfg['K'], cfg['P_sum'], dev, T, cfg)
    return _store(record_trajectories(_load(diffusion_model, ckpt_path, dev), X_test, omega), out_csv, log)


def sum_rate_dataset_store(out_csv=None, sample_num=2000, M=80, W=20.0, g_range=(0.5, 2.5), gs=None, log=print):
    """datasets/sum_rate_gen.py:9-14: `[gs (M) | rate (1) | schemes (M)]` per row, no header - the layout `msr_data_load`
    reads back (MSR.py:164-170).  Labels come from the device label generator (labelgen.SUM_RATE_GEN)."""
    from .labelgen import SUM_RATE_GEN
    gs, rates, schemes = SUM_RATE_GEN(sample_num=sample_num, M=M, g_range=g_range, W=W, gs=gs)
    table = np.concatenate((gs, np.atleast_2d(rates).T, schemes), axis=1)
    if out_csv is None:
        out_csv = f"../datasets/{M}c_{int(W)}w.csv"
    pd.DataFrame(table).to_csv(out_csv, index=False, header=None)
    log("Data generation finished.")
    return table
