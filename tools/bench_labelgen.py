#!/usr/bin/env python3
"""MSR label generator: device time for the reference's job (datasets/sum_rate_gen.py: 2000 x 80, W = 20)
and for 65 536 instances, next to the CPU restatement on a bounded sample.  python tools/bench_labelgen.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from diffsg_amd.labelgen import SUM_RATE_GEN
from oracle import sumrate_oracle as S
out = {}
rng = np.random.default_rng(0)
for rows in (2000, 65536):
    gs = rng.uniform(0.5, 2.5, size=(rows, 80))
    SUM_RATE_GEN(sample_num=rows, M=80, W=20.0, gs=gs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): SUM_RATE_GEN(sample_num=rows, M=80, W=20.0, gs=gs)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    # algorithmic work: 149 iterations x M^2 compare-adds per instance (float64)
    out[str(rows)] = {"ms_per_call_incl_copies": ms, "instances_per_s": rows / ms * 1e3, "gcompare_adds_per_s": 149 * 80 * 80 * rows / ms / 1e6}
t0 = time.perf_counter(); S.sum_rate_gen(rng.uniform(0.5, 2.5, size=(64, 80)), 20.0); cpu = time.perf_counter() - t0
out["cpu_oracle"] = {"instances": 64, "seconds": cpu, "instances_per_s": 64 / cpu}
print(json.dumps(out))
