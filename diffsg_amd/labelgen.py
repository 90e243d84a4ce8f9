"""Label generators on the device: SUM_RATE_GEN (MSR) and CONV_CO_MINLP_GEN (CO).

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


def range_random(mu, sigma, size, lower=None, upper=None):
    """utils/dataset_generate.py:5-24: normal draws, out-of-range entries re-drawn until none is left (numpy's global
    generator, same calls in the same order as the reference)."""
    arr = np.random.normal(mu, sigma, size)
    if lower is None or upper is None:
        return arr
    while np.any(arr < lower) or np.any(arr > upper):
        arr[arr < lower] = np.random.normal(mu, sigma, np.sum(arr < lower))
        arr[arr > upper] = np.random.normal(mu, sigma, np.sum(arr > upper))
    return arr


def CONV_CO_MINLP_GEN(node_num, sample_num, step=0.02, device=None, log=print):
    """utils/dataset_generate.py:147-245: labels of the conventional computation-offloading MINLP by exhaustive search
    (2^n decisions x the `step` allocation grid).  Same signature, draws and return value as the reference -- X
    [samples][6n + 7] features, Y [samples][2n + 1] = decision | allocation | cost -- and the same two report lines; the
    draws and the derived per-node quantities are numpy on the host (:169-184), the search runs in libdiffsg_hip.so
    (csrc/dsg_cogen.hpp, one workgroup per sample; the reference spends ~1 s per 3-node sample on it)."""
    import time
    if not torch.cuda.is_available():
        raise RuntimeError("CONV_CO_MINLP_GEN: no HIP device; libdiffsg_hip has no CPU path")
    F_t, kappa, P_t, P_I, theta, B, N0 = 2.5e9, 1e-28, 0.3, 0.1, 1.0, 10e5, 7.96159e-13
    n = int(node_num)
    params = np.empty((sample_num, 7, n), dtype=np.float64)
    X = np.empty((sample_num, 6 * n + 7), dtype=np.float64)
    for i in range(sample_num):
        s = range_random(2.5e5, 5e4, n, 0, 5e5).astype(int)
        c = s * 3e3
        f_local = range_random(5.0e8, 2.0e8, n, 0, 1e9).astype(int)
        alpha = np.random.rand(n)
        beta = 1 - alpha
        h = np.random.rand(n)
        sinr = P_t * (h ** 2) / (N0 + np.sum(P_t * (h ** 2)))
        r_u = B * np.log2(1 + sinr)
        cost_local = alpha * (c / f_local) + beta * (kappa * (f_local ** 2) * c)
        params[i] = (s, c, f_local, alpha, beta, r_u, cost_local)
        X[i, :6 * n] = np.stack((s, c, f_local, h, alpha, beta), axis=1).reshape(-1)
        X[i, 6 * n:] = (F_t, kappa, P_t, P_I, theta, B, N0)
    dev = torch.device(device if device is not None else "cuda")
    choices = np.arange(step, 1 + step, step)
    t0 = time.time()
    P = torch.from_numpy(params).to(dev)
    ch = torch.from_numpy(choices).to(dev)
    Y = torch.empty(sample_num, 2 * n + 1, device=dev, dtype=torch.float64)
    tol = torch.empty(sample_num, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().dsg_co_minlp_search(_lib.ptr(P), _lib.ptr(ch), len(choices), _lib.ptr(Y), _lib.ptr(tol), sample_num, n,
                                                  F_t, P_t, P_I, theta, _lib.stream_ptr()))
    Yh, hits = Y.cpu().numpy(), int(tol.sum().item())
    log(f"{hits}/{sample_num} satisfy the tolerable delay.")
    log(f"{(time.time() - t0) * 1000 / max(sample_num, 1)} ms per sample.")
    return X, Yh
