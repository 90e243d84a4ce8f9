"""CPU restatement of the MSR label generator (test infrastructure: checker only, never the product path).

Reference: utils/dataset_generate.py:247-255 (SUM_RATE_GRAD), :257-278 (alpha_calc), :280-313 (SUM_RATE_GEN) of
qiyu3816/DiffSG -- "LRH gradient descent": every iteration moves power from the channels with the small gradients to the
ones with the large gradients, the total staying W.  float64 numpy, as the reference; vectorised over rows instead of
the reference's Python loops.  Pinned by tests/golden/g8_sum_rate_gen.npz (outputs of the reference functions themselves).
"""
import numpy as np


def sum_rate_grad(gs, schemes):
    """dataset_generate.py:247-255."""
    return gs / ((gs * schemes + 1.0) * np.log(2))


def alpha_calc(grad):
    """dataset_generate.py:257-278.  Walk the channels by decreasing |grad|: +sign until the running sum would reach half
    of the total, the crossing channel takes the fractional remainder, every later channel -sign."""
    ga = np.abs(grad)
    order = np.argsort(-ga, axis=1)
    gs_sorted = np.take_along_axis(ga, order, axis=1)
    sgn = np.where(np.take_along_axis(grad, order, axis=1) > 0, 1.0, -1.0)
    total = np.sum(ga, axis=1)
    alpha_sorted = np.zeros_like(ga)
    for i in range(ga.shape[0]):                      # sequential prefix sums, in the reference's order of additions
        cur, crossed = 0.0, False
        for j in range(ga.shape[1]):
            g = gs_sorted[i, j]
            if crossed:
                alpha_sorted[i, j] = -sgn[i, j]
            elif cur + g >= total[i] / 2:
                alpha_sorted[i, j] = (total[i] - g - 2 * cur) / g * sgn[i, j]
                crossed = True
            else:
                cur = cur + g
                alpha_sorted[i, j] = sgn[i, j]
    alpha = np.zeros_like(ga)
    np.put_along_axis(alpha, order, alpha_sorted, axis=1)
    return alpha


def sum_rate_gen(gs, W, eps=0.001, beta=0.1):
    """dataset_generate.py:280-313 with the channel gains given (the reference draws them itself with np.random.uniform).
    Returns (rates, schemes)."""
    gs = np.asarray(gs, dtype=np.float64)
    n, M = gs.shape
    schemes = np.ones((n, M)) * (W / M)
    k = 1
    grad = sum_rate_grad(gs, schemes)
    while np.any(np.average(np.abs(grad), axis=1) > eps):
        grad = sum_rate_grad(gs, schemes)
        schemes = schemes + beta * alpha_calc(grad) * grad
        k += 1
        if k % 20 == 0:
            beta *= 0.5
        if k == 150:
            break
    return np.sum(np.log2(1.0 + schemes * gs), axis=1), schemes
