"""Solution decoders and objective evaluators of the three problems.

Reference: classifier_free_MSR.py:239-245,287-288; classifier_free_CO.py:255-290; classifier_free_NU.py:267-303.
These run after sampling, once per evaluation, on whatever device the samples live on (device-resident torch
elementwise ops; not part of the timed hot path).
"""
import torch


def msr_decode(y):
    d = (y - y.min()) / (y.max() - y.min())
    return torch.softmax(d, dim=1)


def msr_rate(p_alloc, gains):
    return torch.sum(torch.log2(1.0 + p_alloc * gains), dim=1)


def co_decode(y):
    d = torch.softmax(y, dim=1)
    dead = (y < -10).all(dim=1)
    return torch.where(dead.unsqueeze(1), 0.0, d)


def co_cost(X, Y):
    """Offloaded nodes (Y > 0.1) share the unallocated remainder equally; cost = local | transition + exec / share."""
    n = Y.shape[1]
    D = torch.where(Y > 0.1, 1, 0)
    Y = torch.where(D == 1, Y, 0)
    y_sum = torch.sum(Y, dim=1)
    d_sum = torch.sum(D, dim=1)
    d_sum = torch.where(d_sum == 0, 0.00001, d_sum)
    spread = ((1 - y_sum) / d_sum)[:, None].expand(-1, n)
    Y = torch.where(D == 1, Y + spread, 0.00001)
    local, trans, exe = X[:, 0::3], X[:, 1::3], X[:, 2::3]
    return torch.sum((1 - D) * local + D * (trans + exe / Y), dim=1)


def nu_decode(y, width, height, p_sum):
    d = torch.zeros_like(y)
    lo, hi = torch.min(y[:, :2]), torch.max(y[:, :2])
    d[:, :2] = (y[:, :2] - lo) / (hi - lo)
    d[:, 0] *= width
    d[:, 1] *= height
    d[:, 2:] = torch.softmax(y[:, 2:], dim=1) * p_sum
    return d


def nu_rate(Yd, X):
    """NOMA successive-interference-cancellation rate: users ordered by channel gain (strongest first); the
    strongest sees only noise, user of rank r sees the summed power of ranks < r as interference."""
    sigma_sq, rou_0, H = 110, 60, 150
    K = Yd.shape[1] - 2
    dx = X[:, 0::2] - Yd[:, 0:1]
    dy = X[:, 1::2] - Yd[:, 1:2]
    h = torch.sqrt(rou_0 / (H ** 2 + dx ** 2 + dy ** 2))
    order = torch.argsort(-h, dim=1)
    P = Yd[:, 2:]
    hs = torch.gather(h, 1, order)
    ps = torch.gather(P, 1, order)
    sinr_sorted = torch.zeros_like(ps)
    prev = torch.zeros_like(ps[:, 0])
    for r in range(K):
        if r == 0:
            sinr_sorted[:, 0] = ps[:, 0] * (hs[:, 0] ** 2) / sigma_sq
        else:
            prev = prev + ps[:, r - 1]
            sinr_sorted[:, r] = ps[:, r] / (prev + sigma_sq / (hs[:, r] ** 2))
    return torch.sum(torch.log2(1 + sinr_sorted), dim=1)
