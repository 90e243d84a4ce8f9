#!/usr/bin/env python3
"""Per-parameter gradient error of one training step against the oracle: python tools/grad_err.py [config] [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from _util import synth_params
from oracle import ddpm_oracle as O
from weights import CONFIGS
import test_gpu_parity as TG
name = sys.argv[1] if len(sys.argv) > 1 else "msr3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
plan, p = synth_params(name, 9)
T = 20
ddpm = TG.make_ddpm(name, p, T)
cfg = CONFIGS[name]
g = torch.Generator().manual_seed(B + 1)
y = torch.rand(B, cfg["input_dim"], generator=g); cond = torch.rand(B, cfg["cond_dim"], generator=g)
ts = torch.randint(0, T, (1, B), generator=g); noise = torch.randn(B, cfg["input_dim"], generator=g)
mask = (torch.rand(B, 1, generator=g) < 0.9).float()
bufs = O.schedule_buffers(1.0 - O.cosine_betas(T))
_, ref = O.ddpm_loss_and_grads(p, plan, bufs, T, y, cond, ts, noise, mask)
gmax = max(float(v.abs().max()) for v in ref.values())
first = None
for rep in range(int(os.environ.get('REPS', 2))):
    for q in ddpm.model.parameters(): q.grad = None
    loss = ddpm(y.cuda(), cond.cuda(), ts=ts.cuda(), noise=noise.cuda(), cond_mask=mask.cuda()); loss.backward()
    cur = {k: q.grad.detach().clone() for k, q in ddpm.model.named_parameters()}
    if first is None: first = cur
    else:
        for k in cur:
            if not torch.equal(cur[k], first[k]):
                dd = (cur[k] != first[k]).nonzero()
                print('   NONDET', rep, k, dd.shape[0], 'cols', sorted(set(dd[:, -1].tolist()))[:12])
    rows = []
    for k, prm in ddpm.model.named_parameters():
        got = prm.grad.detach().cpu()
        rows.append((float((got - ref[k]).abs().max()) / gmax, float((got - ref[k]).abs().max()) / float(ref[k].abs().max() + 1e-30), k, tuple(got.shape)))
    rows.sort(reverse=True)
    print("rep", rep, "gmax", gmax)
    for r in rows[:int(os.environ.get('TOP', 8))]: print(f"  {r[0]:.2e} (own {r[1]:.2e}) {r[2]} {r[3]}")
    for r in rows:
        if r[0] > 1e-5:
            d = (dict(ddpm.model.named_parameters())[r[2]].grad.detach().cpu() - ref[r[2]]).abs() / gmax
            bad = (d > 1e-5).nonzero()
            print("   BAD", r[2], "n_bad", bad.shape[0], "rows", sorted(set(bad[:, 0].tolist()))[:40], "cols", sorted(set(bad[:, 1].tolist()))[:40] if d.dim() > 1 else "")
