#!/usr/bin/env python3
"""A few exact-float32 sample() calls (for rocprofv3 --kernel-trace --stats): python tools/f32_trace.py [B] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 20)
ddpm.model.set_precision("f32")
cond = torch.rand(B, 80, device=dev)
ddpm.sample(cond, 1.0, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(calls): ddpm.sample(cond, 1.0, seed=1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / calls / 20
print(f"f32 exact: {dt*1e3:.4f} ms/step = {1/dt:.1f} steps/s")
