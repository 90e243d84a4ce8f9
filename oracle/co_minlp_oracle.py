"""CPU restatement of the CO label generator (test infrastructure: checker only, never the product path).

Reference: utils/dataset_generate.py:147-245 `CONV_CO_MINLP_GEN(node_num, sample_num)` of qiyu3816/DiffSG (+ `range_random`
:5-24, `resource_allocation_gen` :26-49): per sample an exhaustive search over the 2^n offloading decisions D and, for
every D != 0, over all allocations F of the server's capacity in steps of 0.02 to the offloaded nodes with sum(F) = 1
("full" mode, |sum - 1| < 1e-5).  Selection rules, as the reference's loop has them:
  * `optimal`   = the FIRST candidate (enumeration order: D ascending, then allocation index ascending) with the smallest
                  total cost (strict `<`);
  * `tolerable` = the LAST candidate in that order whose delays are all below theta (every hit overwrites the previous
                  one -- not the cheapest); if any exists it replaces the optimum.
float64 numpy with the reference's expressions term for term (so results are bit-identical); vectorised over the
allocations of one decision instead of the reference's Python loop.  Pinned by tests/golden/g11_co_minlp.npz.
"""
import numpy as np

F_T, KAPPA, P_T, P_I, THETA, BW, N0 = 2.5e9, 1e-28, 0.3, 0.1, 1.0, 10e5, 7.96159e-13


def range_random(mu, sigma, size, lower=None, upper=None):
    """dataset_generate.py:5-24 (same numpy calls in the same order: it consumes the global generator like the reference)."""
    arr = np.random.normal(mu, sigma, size)
    if lower is None or upper is None:
        return arr
    while np.any(arr < lower) or np.any(arr > upper):
        arr[arr < lower] = np.random.normal(mu, sigma, np.sum(arr < lower))
        arr[arr > upper] = np.random.normal(mu, sigma, np.sum(arr > upper))
    return arr


def draw_sample(node_num):
    """The random draws of one sample, dataset_generate.py:169-176, in the reference's order."""
    s = range_random(2.5e5, 5e4, node_num, 0, 5e5).astype(int)
    f_local = range_random(5.0e8, 2.0e8, node_num, 0, 1e9).astype(int)
    alpha = np.random.rand(node_num)
    h = np.random.rand(node_num)
    return s, f_local, alpha, h


def choices(step=0.02):
    """dataset_generate.py:32: the allocation grid (np.arange with a float step: 50 values, the last one 1.0000000000000002)."""
    return np.arange(step, 1 + step, step)


def derived(s, f_local, alpha, h):
    """dataset_generate.py:170-184: c, beta, r_u, cost_local."""
    c = s * 3e3
    beta = 1 - alpha
    sinr = P_T * (h ** 2) / (N0 + np.sum(P_T * (h ** 2)))
    r_u = BW * np.log2(1 + sinr)
    tau_local = c / f_local
    epsilon_local = KAPPA * (f_local ** 2) * c
    cost_local = alpha * tau_local + beta * epsilon_local
    return c, beta, r_u, cost_local


def allocations(D, ch):
    """resource_allocation_gen(D, 'full', step), dataset_generate.py:26-49: rows in the reference's enumeration order."""
    idx = np.where(D == 1)[0]
    n_comb = len(ch) ** len(idx)
    i = np.arange(n_comb)
    arrays = np.zeros((n_comb, len(D)))
    for j, k in enumerate(idx):
        arrays[:, k] = ch[(i // (len(ch) ** j)) % len(ch)]
    return arrays[np.abs(np.sum(arrays, axis=-1) - 1) < 10e-6]


def solve(s, f_local, alpha, h, step=0.02):
    """One sample: returns (x row [6n + 7], y row [2n + 1] = D | F | cost, found_tolerable)."""
    n = len(s)
    c, beta, r_u, cost_local = derived(s, f_local, alpha, h)
    ch = choices(step)
    best = (np.inf, None, None)
    tol = None
    for d in range(2 ** n):
        D = np.array([(d >> j) & 1 for j in range(n)])
        Fs = np.atleast_2d(np.zeros(n)) if d == 0 else allocations(D, ch)
        if Fs.shape[0] == 0:
            continue
        F = np.where(D > 0, Fs, 0.00001)
        cost_off = np.where(D > 0, alpha * (s / r_u + c / (F_T * F)) + beta * (P_T * s / r_u + P_I * c / (F_T * F)), 0)
        delays = np.where(D > 0, s / r_u + c / (F_T * F), c / f_local)
        ok = np.all(delays < THETA, axis=1)
        cost = np.sum((1 - D) * cost_local + D * cost_off, axis=1)
        Fz = np.where(D > 0, F, 0)
        k = int(np.argmin(cost))                     # first occurrence = the reference's strict `<` update
        if cost[k] < best[0]:
            best = (cost[k], D, Fz[k])
        if ok.any():
            k = int(np.nonzero(ok)[0][-1])           # every tolerable candidate overwrites: the last one stays
            tol = (cost[k], D, Fz[k])
    x = np.concatenate([np.array([s[i], c[i], f_local[i], h[i], alpha[i], beta[i]], dtype=np.float64) for i in range(n)]
                       + [np.array([F_T, KAPPA, P_T, P_I, THETA, BW, N0])])
    cost_, D_, F_ = tol if tol is not None else best
    return x, np.concatenate((D_.astype(np.float64), F_, [cost_])), tol is not None


def conv_co_minlp_gen(node_num, sample_num, step=0.02):
    """CONV_CO_MINLP_GEN with numpy's global generator, as the reference: (X [samples][6n+7], Y [samples][2n+1], hits)."""
    X, Y, hits = [], [], 0
    for _ in range(sample_num):
        x, y, t = solve(*draw_sample(node_num), step=step)
        X.append(x); Y.append(y); hits += int(t)
    return np.array(X), np.array(Y), hits
