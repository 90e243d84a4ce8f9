"""world_size-2 `gloo` tests of the multi-GPU host logic on CPU: row sharding of reverse sampling (no collective in
the sampling path), the single flat-bucket gradient all-reduce of data-parallel training, and weight broadcast.
The HIP kernels themselves cannot run here; the compute entry points are replaced by deterministic stand-ins so
that what is tested is exactly the distributed plumbing bench.py and train.py use."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from diffsg_amd import UNet1D, generate_cosine_schedule
        from diffsg_amd.classifier_free_MSR import DDPM
        from diffsg_amd import parallel as par

        # ---- row sharding: balanced, contiguous, covers every row once
        for n in (1, 2, 7, 512, 65537):
            spans = [par.shard_rows(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1

        torch.manual_seed(rank)  # ranks start from different weights
        m = UNet1D(input_dim=3, proj_dim=16, cond_dim=3, dims=(16, 8, 4), is_attn=(False,) * 3, n_blocks=2)
        d = DDPM(20, m, 3, 10.0, 1.0 - generate_cosine_schedule(20), torch.device("cpu"), (1, 3), None)
        par.broadcast_parameters(d.model, src=0)
        ref = [torch.zeros_like(p) for p in d.model.parameters()]
        for r, p in zip(ref, d.model.parameters()):
            r.copy_(p.data)
            dist.broadcast(r, 0)
            assert torch.equal(r, p.data)

        # ---- sampling: every rank handles its own shard; gather only for the check.  Stand-in sampler = row-local map.
        class Fake:
            def sample(self, cond, omega=1.0, **kw):
                return cond * 2.0 + omega
        cond_all = torch.arange(11 * 3, dtype=torch.float32).reshape(11, 3)
        y = par.sample_sharded(Fake(), cond_all, 0.5, gather=True)
        assert torch.equal(y, cond_all * 2.0 + 0.5)
        lo, hi = par.shard_rows(11, rank, ws)
        assert par.sample_sharded(Fake(), cond_all, 0.5).shape[0] == hi - lo

        # ---- training: ONE all-reduce over the flat bucket gives the mean of the ranks' gradients in every .grad view
        total = sum(p.numel() for p in d.model.parameters())
        d._grad_pool = []
        d._grad_bucket = torch.zeros(total)
        d._publish(torch.tensor(1.0), torch.full((total,), float(rank + 1)))
        calls = []
        orig = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        d.allreduce_grads()
        dist.all_reduce = orig
        assert len(calls) == 1
        want = sum(range(1, ws + 1)) / ws
        assert all(torch.all(p.grad == want) for p in d.model.parameters())
        assert all(p.grad is None for p in d.ema.parameters())
        assert d.model.feature_proj.weight.grad.data_ptr() == d.grad_bucket.data_ptr()
        # a second backward without zero_grad accumulates, as autograd would
        d._publish(torch.tensor(2.0), torch.full((total,), 1.0))
        assert torch.all(d.model.final.bias.grad == want + 2.0)
        # ---- the train entry points' data-parallel plumbing (train.dp_context / make_loader / sync_replicas / run_epochs) with a
        # stand-in for the fused HIP step: replicas that start from different weights and see different row shards must hold
        # bit-identical weights after two steps, and only rank 0 reports
        from diffsg_amd import train as tr
        from diffsg_amd.ddpm import _PublishGrads
        dev_, rank_, world_ = tr.dp_context()
        assert (rank_, world_) == (rank, ws) and dev_ is None        # no GPU here: the entry points raise at this point
        assert torch.initial_seed() != 0 or rank == 0                # per-rank seed offset applied
        torch.manual_seed(100 + rank)
        m2 = UNet1D(input_dim=3, proj_dim=16, cond_dim=3, dims=(16, 8, 4), is_attn=(False,) * 3, n_blocks=2)
        d2 = DDPM(20, m2, 3, 10.0, 1.0 - generate_cosine_schedule(20), torch.device("cpu"), (1, 3), None)
        w0 = d2.model.final.weight.detach().clone()
        tr.sync_replicas(d2)
        ws_ = [torch.zeros_like(w0) for _ in range(ws)]
        dist.all_gather(ws_, d2.model.final.weight.detach().clone())
        assert torch.equal(ws_[0], ws_[1]) and (rank == 0) == torch.equal(ws_[0], w0)
        total2 = sum(p.numel() for p in d2.model.parameters())
        seen = []

        def fake_forward(y, cond):                                    # loss and "gradients" that depend on this rank's rows
            seen.append(y[:, 0].clone())
            if getattr(d2, "_grad_bucket", None) is None:
                d2._grad_pool = []
                d2._grad_bucket = torch.zeros(total2)
                d2._loss_anchor = torch.zeros((), requires_grad=True)
            return _PublishGrads.apply(d2._loss_anchor, y.mean().detach(), d2, torch.full((total2,), float(y.mean())))
        d2.forward = fake_forward
        rows = torch.arange(40, dtype=torch.float32)[:, None].repeat(1, 3)
        ds = torch.utils.data.TensorDataset(rows.clone(), rows.clone())
        loader = tr.make_loader(ds, 10, rank, ws, seed=3)
        opt = tr.FlatAdam(d2, lr=1e-2, fused=False)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, [5])
        logs = []
        before = d2.model.final.weight.detach().clone()
        tr.run_epochs(d2, loader, opt, sched, 1, False, 5, torch.device("cpu"), logs.append)
        assert len(seen) == 2 and (len(logs) == 1) == (rank == 0)
        mine = torch.cat(seen)
        both = [torch.zeros_like(mine) for _ in range(ws)]
        dist.all_gather(both, mine)
        assert sorted(torch.cat(both).tolist()) == list(map(float, range(40)))   # the ranks' shards partition the rows
        after = d2.model.final.weight.detach().clone()
        dist.all_gather(ws_, after)
        assert torch.equal(ws_[0], ws_[1]) and not torch.equal(after, before)
        # ---- healthy path: per step exactly ONE gradient-bucket all-reduce, plus step_health's two scalars
        sizes_seen = []
        orig_ar = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (sizes_seen.append(t.numel()), orig_ar(t, *a, **k))[1]
        tr.run_epochs(d2, loader, opt, sched, 1, False, 5, torch.device("cpu"), logs.append)
        dist.all_reduce = orig_ar
        assert sizes_seen == [total2, 2, total2, 2], sizes_seen
        # ---- a failing rank fails the JOB, in the same step, instead of leaving the others in the next gradient all-reduce: rank 1 reports
        # a set fp16 range flag -> both ranks raise FloatingPointError, each after exactly one step
        seen.clear()
        d2.model.range_exceeded = lambda: rank == 1
        try:
            tr.run_epochs(d2, loader, opt, sched, 1, False, 5, torch.device("cpu"), logs.append)
            raise AssertionError("run_epochs returned although rank 1 reported a set range flag")
        except FloatingPointError as e:
            assert len(seen) == 1 and ("this rank" in str(e)) == (rank == 1) and "fp16 range" in str(e), str(e)
        del d2.model.range_exceeded
        # ... and a non-finite loss on one rank
        seen.clear()
        real_forward = d2.forward

        def nan_forward(y, cond):
            out = real_forward(y, cond)
            return out * float("nan") if rank == 0 else out
        d2.forward = nan_forward
        try:
            tr.run_epochs(d2, loader, opt, sched, 1, False, 5, torch.device("cpu"), logs.append)
            raise AssertionError("run_epochs returned although rank 0's loss was NaN")
        except FloatingPointError as e:
            assert len(seen) == 1 and "not finite" in str(e), str(e)
        d2.forward = real_forward
        # ---- sharded sampling with collectives inside or behind the call: a rank whose sample() raises must not leave the others waiting
        class Failing:
            T = 20

            def sample(self, cond, omega=1.0, **kw):
                gr = getattr(self, "gr", None)
                for k in range(4):                   # the library calls the hook on each of the min(T, 4) early steps
                    if gr is not None:
                        gr.stats.fill_(1.0)
                        gr._callback(None)
                    if rank == 1 and k == 0:
                        raise ValueError("rank 1 fails after its first renorm reduction")
                return cond * 2.0
        fail = Failing()
        for kw in ({"gather": True}, {"global_renorm_stats": True}, {"global_renorm_stats": True, "gather": True}):
            class FakeRenorm(par.global_renorm):     # the real bookkeeping (n_reduced, contribute_remaining, check) without the library
                def __enter__(self):
                    self.stats = torch.zeros(3, dtype=torch.float64)
                    self.n_reduced = 0
                    fail.gr = self if "global_renorm_stats" in kw else None
                    return self

                def __exit__(self, *exc):
                    fail.gr = None
                    return False
            real_gr, par.global_renorm = par.global_renorm, FakeRenorm
            try:
                par.sample_sharded(fail, cond_all, 0.5, **kw)
                raise AssertionError(f"sample_sharded({kw}) returned although rank 1 failed")
            except ValueError as e:
                assert rank == 1 and "rank 1 fails" in str(e)
            except RuntimeError as e:
                assert rank == 0 and "another rank" in str(e), str(e)
            finally:
                par.global_renorm = real_gr
        dist.barrier()                               # both ranks are past every collective of the failed calls: nothing is left pending
        # ---- the evidence a multi-rank bench line carries (bench.py): every rank counted, per-rank rates gathered, and the
        # post-all-reduce bucket identical bit for bit on every rank -- and the check FAILS when one rank's bucket differs
        ev = par.run_evidence(torch.device("cpu"), 0.5 + 0.25 * rank, 10)
        assert ev["ranks_seen"] == ws and len(ev["per_rank_steps_per_s"]) == ws
        assert ev["max_seconds"] == 0.5 + 0.25 * (ws - 1) and ev["per_rank_steps_per_s"][0] == 20.0
        eq, csum = par.bucket_checksum_equal(d.grad_bucket)
        assert eq and len(csum) == 2
        skew = d.grad_bucket.clone()
        if rank == 1:
            skew[7] = torch.nextafter(skew[7], torch.tensor(1e9))     # one ulp on one rank
        eq2, _ = par.bucket_checksum_equal(skew)
        assert not eq2
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gloo_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(r for r, _ in res) == [0, 1]
    for r, msg in res:
        assert msg == "ok", f"rank {r}:\n{msg}"
