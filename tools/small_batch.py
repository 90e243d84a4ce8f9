#!/usr/bin/env python3
"""One reference-shaped evaluation call (B=512, T=20, omega=500) repeated, for kernel-trace timelines."""
import os, sys, time
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
if os.environ.get("TILE") is not None:      # 0: one launch per operator (tile_step off); 1: k_unet_tile (default)
    d.model.set_option("tile_step", int(os.environ["TILE"]))
for _ in range(3):
    d.sample(cond, 500.0, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    d.sample(cond, 500.0, seed=2)
torch.cuda.synchronize()
print(f"{name} B={B} T={T} tile_step={os.environ.get('TILE', 'default')}: {(time.perf_counter()-t0)/10*1e3:.3f} ms per sample() call")
