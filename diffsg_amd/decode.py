"""Solution decoders and objective evaluators of the three problems, on the device.

Reference: classifier_free_MSR.py:239-245,287-288; classifier_free_CO.py:255-290; classifier_free_NU.py:267-303.
Every function is one call into libdiffsg_hip.so (csrc/dsg_eval.hpp) on the tensors' device and stream; like the rest of
the package there is no CPU path (the CPU restatement used by the tests lives in oracle/).
"""
import torch

from . import _lib


def _dev(*ts):
    out = []
    for t in ts:
        if not torch.is_tensor(t):
            t = torch.as_tensor(t)
        if not t.is_cuda:
            raise RuntimeError("diffsg_amd.decode: tensors are not on a HIP device; libdiffsg_hip has no CPU path")
        out.append(t.detach().to(torch.float32).contiguous())
    if any(t.device != out[0].device for t in out):
        raise RuntimeError("diffsg_amd.decode: tensors live on different devices")
    return out


def _rows2d(t, what):
    if t.dim() != 2:
        raise ValueError(f"{what}: expected a (rows, columns) tensor, got {tuple(t.shape)}")
    return t.shape[0], t.shape[1]


def _call(name, dev, *args):
    with torch.cuda.device(dev):
        _lib.check(getattr(_lib.lib(), name)(*args, _lib.stream_ptr()))


def row_softmax(y):
    (y,) = _dev(y)
    rows, D = _rows2d(y, "row_softmax")
    out = torch.empty_like(y)
    _call("dsg_row_softmax", y.device, _lib.ptr(y), _lib.ptr(out), rows, D)
    return out


def msr_decode(y):
    """custom_decoder, classifier_free_MSR.py:239-245: min-max over the whole tensor, then a row softmax."""
    (y,) = _dev(y)
    rows, D = _rows2d(y, "msr_decode")
    out = torch.empty_like(y)
    _call("dsg_msr_decode", y.device, _lib.ptr(y), _lib.ptr(out), rows, D)
    return out


def msr_rate(p_alloc, gains):
    """classifier_free_MSR.py:287-288: sum_c log2(1 + p * gain)."""
    p_alloc, gains = _dev(p_alloc, gains)
    rows, D = _rows2d(p_alloc, "msr_rate")
    if gains.shape != p_alloc.shape:
        raise ValueError("msr_rate: allocation and gains differ in shape")
    out = torch.empty(rows, device=p_alloc.device, dtype=torch.float32)
    _call("dsg_msr_rate", p_alloc.device, _lib.ptr(p_alloc), _lib.ptr(gains), _lib.ptr(out), rows, D)
    return out


def co_decode(y):
    """customized_real_decoder, classifier_free_CO.py:281-290: row softmax; rows with every entry < -10 become zero."""
    (y,) = _dev(y)
    rows, D = _rows2d(y, "co_decode")
    out = torch.empty_like(y)
    _call("dsg_co_decode", y.device, _lib.ptr(y), _lib.ptr(out), rows, D)
    return out


def co_cost(X, Y):
    """cost_calc, classifier_free_CO.py:255-278: offloaded nodes (Y > 0.1) share the unallocated remainder equally;
    cost = local | transition + exec / share."""
    X, Y = _dev(X, Y)
    rows, n = _rows2d(Y, "co_cost")
    if X.shape != (rows, 3 * n):
        raise ValueError(f"co_cost: X is {tuple(X.shape)}, expected {(rows, 3 * n)}")
    out = torch.empty(rows, device=Y.device, dtype=torch.float32)
    _call("dsg_co_cost", Y.device, _lib.ptr(X), _lib.ptr(Y), _lib.ptr(out), rows, n)
    return out


def nu_decode(y, width, height, p_sum):
    """custom_decoder, classifier_free_NU.py:267-276."""
    (y,) = _dev(y)
    rows, D = _rows2d(y, "nu_decode")
    out = torch.empty_like(y)
    import ctypes
    _call("dsg_nu_decode", y.device, _lib.ptr(y), _lib.ptr(out), rows, D, ctypes.c_float(width), ctypes.c_float(height),
          ctypes.c_float(p_sum))
    return out


def nu_rate(Yd, X):
    """rate_calc, classifier_free_NU.py:279-303: NOMA successive-interference-cancellation sum rate (users ordered by
    channel gain, strongest first; rank r sees the summed power of ranks < r)."""
    Yd, X = _dev(Yd, X)
    rows, D = _rows2d(Yd, "nu_rate")
    K = D - 2
    if X.shape != (rows, 2 * K):
        raise ValueError(f"nu_rate: X is {tuple(X.shape)}, expected {(rows, 2 * K)}")
    out = torch.empty(rows, device=Yd.device, dtype=torch.float32)
    _call("dsg_nu_rate", Yd.device, _lib.ptr(Yd), _lib.ptr(X), _lib.ptr(out), rows, K)
    return out
