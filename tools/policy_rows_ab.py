#!/usr/bin/env python3
"""Sampling step time by batch size under launch policies (which form wins where): python tools/policy_rows_ab.py [config]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from weights import CONFIGS
from diffsg_amd import UNet1D, generate_cosine_schedule, init_weights
from diffsg_amd.classifier_free_MSR import DDPM
name = sys.argv[1] if len(sys.argv) > 1 else "msr80"
T = 20
dev = torch.device("cuda:0")
cfg = CONFIGS[name]
torch.manual_seed(0)
m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
D = cfg["input_dim"]
d = DDPM(T, m, D, 10.0, 1.0 - generate_cosine_schedule(T), dev, (1, D), None)
d.apply(init_weights); d.to(dev)
for B in (4096, 8192, 12288, 16384, 24576, 32768):
    cond = torch.rand(B, cfg["cond_dim"], device=dev)
    row = []
    for pol in ((-1, -1), (1024, -1), (2048, -1), (1024, 2048), (2048, 2048)):
        d.model.set_launch_policy(*pol)
        for _ in range(2): d.sample(cond, 1.0, seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): d.sample(cond, 1.0, seed=1)
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / 5 / T * 1e3)
    print(f"{name} B={B:6d} ({2 * ((B + 31) // 32)} tiles): ms/step default {row[0]:.4f} | coop<=1024 {row[1]:.4f} | coop<=2048 {row[2]:.4f} | "
          f"coop<=1024,narrow_small<=2048 {row[3]:.4f} | coop<=2048,narrow_small<=2048 {row[4]:.4f}", flush=True)
