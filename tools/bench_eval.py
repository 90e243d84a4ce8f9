#!/usr/bin/env python3
"""Decoders / evaluators against the HBM roofline: python tools/bench_eval.py [rows]
Prints one JSON object: per kernel the algorithmic bytes (inputs read once + outputs written once), the time per call
(HIP events, 50 calls) and the achieved fraction of 8 TB/s; plus the CPU restatement (oracle) timed on the host."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffsg_amd import decode as Dc
from oracle import ddpm_oracle as O
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
y = torch.randn(rows, 80, generator=g); gain = torch.rand(rows, 80, generator=g) * 5
yc = torch.randn(rows, 3, generator=g); Xc = torch.rand(rows, 9, generator=g) * 10
yn = torch.randn(rows, 5, generator=g); Xn = torch.rand(rows, 6, generator=g) * 400
Y, G, YC, XC, YN, XN = (t.to(dev) for t in (y, gain, yc, Xc, yn, Xn))
dec, dco, dnu = Dc.msr_decode(Y), Dc.co_decode(YC), Dc.nu_decode(YN, 400, 400, 18.0)
cases = [
    ("dsg_msr_decode", lambda: Dc.msr_decode(Y), (2 * 80 + 80) * 4 * rows, lambda: O.msr_decode(y)),   # read twice: global min-max pass + softmax pass
    ("dsg_msr_rate", lambda: Dc.msr_rate(dec, G), (160 + 1) * 4 * rows, lambda: O.msr_rate(O.msr_decode(y), gain)),
    ("dsg_co_decode", lambda: Dc.co_decode(YC), 6 * 4 * rows, lambda: O.co_decode(yc)),
    ("dsg_co_cost", lambda: Dc.co_cost(XC, dco), 13 * 4 * rows, lambda: O.co_cost(Xc, O.co_decode(yc))),
    ("dsg_nu_decode", lambda: Dc.nu_decode(YN, 400, 400, 18.0), (5 + 2 + 5) * 4 * rows, lambda: O.nu_decode(yn, 400, 400, 18.0)),
    ("dsg_nu_rate", lambda: Dc.nu_rate(dnu, XN), 12 * 4 * rows, lambda: O.nu_rate(O.nu_decode(yn, 400, 400, 18.0), Xn)),
]
out = {"rows": rows, "peak_gbs": 8000.0, "kernels": {}}
for name, fn, nbytes, cpu in cases:
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    t0 = time.perf_counter(); cpu(); cpu_ms = (time.perf_counter() - t0) * 1e3
    out["kernels"][name] = {"ms_per_call": ms, "algorithmic_bytes": nbytes, "achieved_gbs": nbytes / ms / 1e6,
                            "frac_hbm": nbytes / ms / 1e6 / 8000.0, "cpu_oracle_ms": cpu_ms}
print(json.dumps(out))
