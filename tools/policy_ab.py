#!/usr/bin/env python3
"""Sampling step time under launch policies: python tools/policy_ab.py [rows]   (dsg_set_launch_policy: coop_max_tiles, narrow_small_max_tiles)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 20)
cond = torch.rand(B, 80, device=dev)
for pol in ((-1, -1), (1 << 20, -1), (-1, 1 << 20), (1 << 20, 1 << 20), (0, 0)):
    ddpm.model.set_launch_policy(*pol)
    for _ in range(3): ddpm.sample(cond, 1.0, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ddpm.sample(cond, 1.0, seed=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 / 20
    print(f"B={B} policy coop_max={pol[0]} narrow_small_max={pol[1]}: {dt*1e3:.4f} ms/step")
