#!/usr/bin/env python3
"""Round-6 experiment harness for the narrow run: step time and per-operator event times of the reverse step at B rows (default 65536), with a
checksum of the samples (a bit-identical kernel form keeps it).   python tools/narrow_exp.py [rows] [T]
(the DSG_DBG_* variables it prints were read by experiment builds of the library -- wave stagger, workgroup size; profiles/r06_narrow_block_forms.txt)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, T)
cond = torch.rand(B, 80, device=dev)
for _ in range(3): ddpm.sample(cond, 1.0, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): y = ddpm.sample(cond, 1.0, seed=1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4 / T
ddpm.sample(cond, 1.0, seed=1, profile=True); torch.cuda.synchronize()
ops = {r[0][:12]: round(r[3] / T * 1e3, 1) for r in ddpm.op_profile() if r[3] / T > 8e-3}
tag = " ".join(f"{k}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("DSG_DBG"))
print(f"[{tag}] B={B}: {dt*1e3:.4f} ms/step = {1/dt:.1f} steps/s  {ops}  checksum {float(y.double().sum()):.6f}", flush=True)
