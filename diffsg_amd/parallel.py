"""One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Reverse sampling is row-parallel with NO collective: every rank runs `DDPM.sample` on its own row shard, exactly as
the reference treats its 512-row chunks as independent calls (classifier_free_MSR.py:273-279; the early-step global
renorm is per call).  Training is data parallel with ONE all-reduce per step over the flat gradient bucket
(`DDPMCore.allreduce_grads`); `ema.module.*` and the schedule buffers are never communicated.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_rows(n_rows: int, rank: int, world_size: int):
    """Contiguous, balanced [start, stop) of rank's rows (the first n_rows % world ranks get one extra row)."""
    base, extra = divmod(n_rows, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class global_renorm:
    """Context: while active, `ddpm.sample` standardises the early steps with the moments of ALL ranks' rows (one all-reduce
    of 3 float64 scalars on each of the <= 4 renorm steps; dsg_set_renorm_hook) -- the sharded call then reproduces a single
    reference call on the whole batch instead of one call per shard.  `reduce` defaults to torch.distributed.all_reduce."""

    def __init__(self, ddpm, reduce=None):
        from . import _lib
        self.ddpm, self._lib = ddpm, _lib
        self.reduce = reduce if reduce is not None else (lambda t: dist.all_reduce(t))
        self.error = None
        self.n_reduced = 0            # reductions this rank has taken part in since __enter__

    def _callback(self, _user):
        # ctypes swallows an exception raised inside a callback: the rank would silently standardise with its local moments
        # while the others wait in the collective.  Keep it and re-raise once the library call has returned (`check`).
        try:
            self.reduce(self.stats)
            self.n_reduced += 1
        except BaseException as e:  # noqa: BLE001 - re-raised in check()
            if self.error is None:
                self.error = e

    def check(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise RuntimeError("global_renorm: the moment reduction failed inside dsg_sample") from e

    def contribute_nothing(self):
        """A rank whose shard is empty launches nothing, but the other ranks wait in the renorm all-reduces: issue the same
        number of reductions (min(T, 4), MSR.py:136) with zero moments."""
        self.contribute_remaining()

    def contribute_remaining(self):
        """Issue the reductions this rank still owes the group (all of them for an empty shard; the rest of them when its `sample` call
        raised part-way): the other ranks wait in exactly min(T, 4) all-reduces."""
        while self.n_reduced < min(self.ddpm.T, 4):
            self.stats.zero_()
            self.reduce(self.stats)
            self.n_reduced += 1

    def __enter__(self):
        import ctypes
        dev = next(self.ddpm.model.parameters()).device
        self.stats = torch.zeros(3, device=dev, dtype=torch.float64)
        self.n_reduced = 0
        self.cb = self._lib.RENORM_REDUCE_FN(self._callback)
        hd = self.ddpm.model.native_handle()
        self._lib.check(self._lib.lib().dsg_set_renorm_hook(hd, self._lib.ptr(self.stats), ctypes.cast(self.cb, ctypes.c_void_p), None))
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()          # the hook's buffer and callback must outlive the enqueued steps
        self._lib.check(self._lib.lib().dsg_set_renorm_hook(self.ddpm.model.native_handle(), None, None, None))
        if exc[0] is None:
            self.check()
        return False


def sample_sharded(ddpm, cond_all, omega=1.0, gather=False, global_renorm_stats=False, **kw):
    """Each rank samples its shard of `cond_all`.  Default: NO collective -- every shard is its own `sample()` call, exactly
    as the reference treats its 512-row chunks (the early-step renorm is per call).  global_renorm_stats=True: the renorm uses
    the whole batch's statistics (`global_renorm`; 3 scalars all-reduced on 4 steps), matching one reference call on
    `cond_all`.  With gather=True the shards are all-gathered afterwards (a convenience for evaluation, outside the path)."""
    rank, ws = world()
    lo, hi = shard_rows(cond_all.shape[0], rank, ws)
    err, y = None, None
    if global_renorm_stats and ws > 1:
        with global_renorm(ddpm) as gr:
            try:
                y = ddpm.sample(cond_all[lo:hi], omega, **kw)
                if hi == lo:
                    gr.contribute_nothing()
                gr.check()
            except Exception as e:  # noqa: BLE001 - re-raised below, on every rank
                # this rank failed before (or between) its renorm reductions: the others are waiting in them -- pay what is owed, then
                # tell them.  A failure of the reduction itself is not retried: the group is gone.
                err = e
                if gr.error is None:
                    gr.contribute_remaining()
                gr.error = None
    elif ws > 1 and gather:
        try:
            y = ddpm.sample(cond_all[lo:hi], omega, **kw)
        except Exception as e:  # noqa: BLE001 - the other ranks would wait in the all-gather below
            err = e
    else:
        y = ddpm.sample(cond_all[lo:hi], omega, **kw)
    if ws > 1 and (global_renorm_stats or gather):
        raise_everywhere(err, cond_all.device, "sample_sharded")
    if not gather or ws == 1:
        return y
    sizes = [shard_rows(cond_all.shape[0], r, ws) for r in range(ws)]
    width = max(b - a for a, b in sizes)
    pad = torch.zeros(width, y.shape[1], device=y.device, dtype=y.dtype)
    pad[: y.shape[0]] = y
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)])


def raise_everywhere(err, device, where):
    """A call that holds collectives must fail on every rank or on none: all-reduce(MAX) of "this rank failed"; the failing rank
    re-raises its own exception, the others raise a RuntimeError that says so.  No-op when nobody failed."""
    rank, ws = world()
    if ws > 1:
        t = torch.tensor([1.0 if err is not None else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if err is None and t.item() > 0:
            raise RuntimeError(f"{where}: the call failed on another rank of {ws}; this rank's result is discarded")
    if err is not None:
        raise err


def broadcast_parameters(module, src=0):
    """Replicas start from rank `src`'s weights (one flat broadcast per dtype group is not needed at 6 MB)."""
    rank, ws = world()
    if ws == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def run_evidence(device, seconds_this_rank, steps):
    """What a multi-rank bench line needs to prove itself on hardware nobody has run it on before: `ranks_seen` (an all-reduced
    count: every rank of the launch took part in a collective), the per-rank step rates (an all-gather of each rank's own
    wall time for the same `steps`), and the slowest rank's time.  One process: the trivial record."""
    rank, ws = world()
    if ws == 1:
        return {"ranks_seen": 1, "per_rank_steps_per_s": [steps / seconds_this_rank], "max_seconds": seconds_this_rank}
    one = torch.ones(1, device=device, dtype=torch.float64)
    dist.all_reduce(one)
    t = torch.tensor([seconds_this_rank], device=device, dtype=torch.float64)
    parts = [torch.zeros_like(t) for _ in range(ws)]
    dist.all_gather(parts, t)
    secs = [float(x.item()) for x in parts]
    return {"ranks_seen": int(round(float(one.item()))), "per_rank_steps_per_s": [steps / x for x in secs], "max_seconds": max(secs)}


def bucket_checksum_equal(bucket):
    """After the gradient all-reduce every rank must hold the SAME bucket, bit for bit (same reduction result delivered to
    everyone): a float64 sum and a wrap-around sum of the bit patterns, all-gathered and compared.  Returns (equal, checksum)."""
    rank, ws = world()
    bits = bucket.detach().view(torch.int32).to(torch.int64).sum()
    val = bucket.detach().double().sum()
    mine = torch.stack([bits.double(), val]).to(torch.float64)
    if ws == 1:
        return True, [float(mine[0]), float(mine[1])]
    parts = [torch.zeros_like(mine) for _ in range(ws)]
    dist.all_gather(parts, mine)
    return all(torch.equal(p, parts[0]) for p in parts), [float(parts[0][0]), float(parts[0][1])]
