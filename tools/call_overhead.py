#!/usr/bin/env python3
"""Per-call overhead of DDPM.sample at the bench workload: time(T) = a + b*T.  python tools/call_overhead.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
cond = torch.rand(B, 80, device=dev)
res = {}
for T in (1, 2, 5, 10, 20, 50, 100):
    ddpm = bench.build_model(dev, T)
    ddpm.sample(cond, 1.0, seed=1); torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ddpm.sample(cond, 1.0, seed=2 + rep)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res[T] = min(ts) * 1e3
    print(f"T={T:4d}  {res[T]:8.3f} ms/call  {res[T]/T:7.3f} ms/step")
b = (res[100] - res[20]) / 80; a = res[20] - 20 * b
print(f"fit: {a:.3f} ms/call + {b:.4f} ms/step")
