#!/usr/bin/env python3
"""Host time per training step (32 768 rows, MSR-80c): how long the host needs to ENQUEUE a step -- eager loop vs the captured step graph
(train.StepGraph) -- and the steady-state step time of both.   python tools/train_host_prof.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from diffsg_amd.train import FlatAdam, StepGraph
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 20)
opt = FlatAdam(ddpm, lr=0.005)
ddpm.device_draws = 1
cond = torch.rand(B, 80, device=dev); y = torch.rand(B, 80, device=dev) * 0.25
def one():
    loss = ddpm(y, cond); loss.backward(); opt.step(); opt.zero_grad(); return loss
for _ in range(6): one()
def measure(fn, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    return t_host * 1e3, t_all * 1e3
h, a = measure(one)
print(f"eager: host {h:.3f} ms per step to enqueue, {a:.3f} ms per step in steady state (B={B})")
sg = StepGraph(ddpm, opt, y, cond)
for _ in range(3): sg.step()
h, a = measure(sg.step)
print(f"graph: host {h:.3f} ms per step to enqueue, {a:.3f} ms per step in steady state (B={B})")
