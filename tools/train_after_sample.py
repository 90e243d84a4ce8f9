#!/usr/bin/env python3
"""Does a sampling call earlier in the process change the training step's time?  python tools/train_after_sample.py [sample_first 0|1]
(bench.py's train leg runs behind its sampling leg in one process; tools/train_ab.py trains in a fresh process.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from diffsg_amd.train import FlatAdam
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
if first:
    s = bench.build_model(dev, 20)
    cond = torch.rand(65536, 80, device=dev)
    for _ in range(3): s.sample(cond, 1.0, seed=1)
    torch.cuda.synchronize()
    if first == 2:
        del s, cond
        torch.cuda.empty_cache()
B, steps = 32768, 30
ddpm = bench.build_model(dev, 20)
FlatAdam.native_step = True; ddpm.device_draws = 1000
opt = FlatAdam(ddpm, lr=0.005)
cond = torch.rand(B, 80, device=dev); y = torch.rand(B, 80, device=dev) * 0.25
def one():
    loss = ddpm(y, cond); loss.backward(); ddpm.allreduce_grads(); opt.step(); opt.zero_grad(); return loss
for v in (1, 0, 1, 0):
    ddpm.model.set_option("train_time_beside", v)
    for _ in range(8): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"sample_first={first} train_time_beside={v}: {dt*1e3:.3f} ms/step", flush=True)
