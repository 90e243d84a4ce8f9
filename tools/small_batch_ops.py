#!/usr/bin/env python3
"""Per-operator time of one reverse step at small batch (HIP events, eager): python tools/small_batch_ops.py [config] [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from weights import CONFIGS
from diffsg_amd import UNet1D, generate_cosine_schedule, init_weights
from diffsg_amd.classifier_free_MSR import DDPM
name = sys.argv[1] if len(sys.argv) > 1 else "msr3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
T = 20
dev = torch.device("cuda:0")
cfg = CONFIGS[name]
torch.manual_seed(0)
m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
D = cfg["input_dim"]
d = DDPM(T, m, D, 10.0, 1.0 - generate_cosine_schedule(T), dev, (1, D), None)
d.apply(init_weights); d.to(dev)
cond = torch.rand(B, cfg["cond_dim"], device=dev)
d.sample(cond, 1.0, seed=1)
d.sample(cond, 1.0, seed=2, profile=True)
torch.cuda.synchronize()
tot = 0.0
for n, fl, by, ms, calls in d.op_profile():
    if calls: print(f"{n:16s} {ms / calls * 1e3:8.1f} us"); tot += ms / calls
print(f"sum {tot * 1e3:.1f} us/step (eager, event overhead ~5 us per operator included)")
