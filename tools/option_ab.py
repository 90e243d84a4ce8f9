#!/usr/bin/env python3
"""Same-box A-B-A-B of a per-handle kernel-form switch on the bench workload:
    python tools/option_ab.py [option=narrow_valu8] [rows=65536] [T=50] [reps=3]
Prints steps/s per setting and round, the per-operator event times of both settings, and the largest relative difference of the results."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
opt = sys.argv[1] if len(sys.argv) > 1 else "narrow_valu8"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
T = int(sys.argv[3]) if len(sys.argv) > 3 else 50
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, T)
cond = torch.rand(B, 80, device=dev)
outs = {}
for rnd in range(reps):
    for val in (0, 1):
        ddpm.model.set_option(opt, val)
        for _ in range(2): ddpm.sample(cond, 1.0, seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): y = ddpm.sample(cond, 1.0, seed=1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4 / T
        outs[val] = y
        print(f"round {rnd} {opt}={val}: {dt*1e3:.4f} ms/step = {1/dt:.1f} steps/s", flush=True)
print("max rel diff between the settings:", float((outs[0] - outs[1]).abs().max() / outs[1].abs().max()))
for val in (0, 1):
    ddpm.model.set_option(opt, val)
    ddpm.sample(cond, 1.0, seed=1, profile=True)
    torch.cuda.synchronize()
    print(f"{opt}={val} per-op ms/step:", {r[0][:24]: round(r[3] / T, 4) for r in ddpm.op_profile() if r[3] / T > 2e-3})
