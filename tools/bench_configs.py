#!/usr/bin/env python3
"""Reverse-sampling throughput on the other BASELINE.json configs (single GPU): steps/s and row-steps/s.
    python tools/bench_configs.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from weights import CONFIGS
from diffsg_amd import UNet1D, generate_cosine_schedule, init_weights
from diffsg_amd.classifier_free_MSR import DDPM

dev = torch.device("cuda:0")
out = []
for name, B, T, omega in (("msr3", 8192, 1000, 1.0), ("msr3", 512, 20, 500.0), ("co3", 8192, 200, 1.0), ("nu3", 8192, 200, 1.0),
                          ("msr80", 8192, 200, 1.0), ("msr80", 512, 20, 500.0)):
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
    D = cfg["input_dim"]
    d = DDPM(T, m, D, 10.0, 1.0 - generate_cosine_schedule(T), dev, (1, D), None)
    d.apply(init_weights)
    d.to(dev)
    cond = torch.rand(B, cfg["cond_dim"], device=dev)
    if os.environ.get("DSG_OPT_V8") is not None:         # A/B of the float32 8-wide section: DSG_OPT_V8=0 / 1
        d.model.set_option("narrow_valu8", int(os.environ["DSG_OPT_V8"]))
    d.sample(cond, omega, seed=1)
    torch.cuda.synchronize()
    dt = 1e9
    for _ in range(3):                       # best of 3 calls
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = d.sample(cond, omega, seed=2)
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    out.append(dict(config=name, B=B, T=T, omega=omega, ms_per_call=dt * 1e3, steps_per_s=T / dt, row_steps_per_s=B * T / dt,
                    finite=bool(torch.isfinite(y).all())))
    print(json.dumps(out[-1]), file=sys.stderr)      # progress, one JSON object per line
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)
print(json.dumps(out))                               # stdout: ONE valid JSON document
