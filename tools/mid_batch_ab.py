#!/usr/bin/env python3
"""Mid-size sampling launches (768 - 2 048 row tiles): the default forms against the persistent panel forms forced by policy (0, x), with
and without the half-panel variant: python tools/mid_batch_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
T = 20
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, T)
for B in (12288, 16384, 24576, 32768, 49152):
    cond = torch.rand(B, 80, device=dev)
    row = []
    for pol, half in (((-1, -1), 0), ((0, 2048), 0), ((0, 2048), 1), ((0, 0), 1)):
        ddpm.model.set_launch_policy(*pol); ddpm.model.set_option("panel_half", half)
        for _ in range(2): ddpm.sample(cond, 1.0, seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ddpm.sample(cond, 1.0, seed=1)
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / 5 / T * 1e3)
    print(f"B={B:6d} ({2 * ((B + 31) // 32)} tiles): ms/step default {row[0]:.4f} | panel forms, small narrow {row[1]:.4f} | + half panels {row[2]:.4f} | half panels, LDS narrow {row[3]:.4f}", flush=True)
