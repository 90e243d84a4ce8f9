#!/usr/bin/env python3
"""ms per reverse step at mid-size batches (the small-launch narrow run up to 2 048 tiles): python tools/ab_mid.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import torch, bench
dev = torch.device("cuda:0")
T = 20
ddpm = bench.build_model(dev, T)
for B in (12288, 16384, 24576, 32768):
    cond = torch.rand(B, 80, device=dev)
    for _ in range(3): ddpm.sample(cond, 1.0, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ddpm.sample(cond, 1.0, seed=1)
    torch.cuda.synchronize(); print(f"B={B}: {(time.perf_counter()-t0)/10/T*1e3:.4f} ms/step", flush=True)
