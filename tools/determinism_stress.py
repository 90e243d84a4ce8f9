#!/usr/bin/env python3
"""Run-to-run determinism stress of the sampling path (all kernel forms): python tools/determinism_stress.py [reps]
Same seed -> the same bits, at the bench size (ring kernels, large narrow run), a cooperative size and a ragged one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_parity import synth_params, make_ddpm, CONFIGS
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for name, B, T in (("msr80", 65536, 20), ("msr80", 4096 + 17, 20), ("co3", 16391, 20), ("nu3", 8192, 20), ("msr80", 512, 20)):
    plan, p = synth_params(name, 4)
    ddpm = make_ddpm(name, p, T)
    cond = torch.rand(B, CONFIGS[name]["cond_dim"], generator=torch.Generator().manual_seed(1)).cuda()
    first = ddpm.sample(cond, 2.0, seed=7).clone()
    diff = 0
    for r in range(reps - 1):
        o = ddpm.sample(cond, 2.0, seed=7)
        if not torch.equal(o, first):
            diff += 1
            d = (o != first).nonzero()
            print(f"  {name} B={B} rep {r + 1}: {d.shape[0]} elements differ, first {d[0].tolist()}, cols {sorted(set(d[:, 1].tolist()))[:16]}")
    print(f"{name} B={B} T={T}: {reps} runs, {diff} differ from the first")
    bad += diff
sys.exit(1 if bad else 0)
