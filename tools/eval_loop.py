#!/usr/bin/env python3
"""The reference's evaluation pattern (load_test_*: the test split in 512-row chunks, one sample() call per chunk,
classifier_free_MSR.py:273-279): wall time per chunk for back-to-back calls.   python tools/eval_loop.py [config] [chunks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from weights import CONFIGS
from diffsg_amd import UNet1D, generate_cosine_schedule, init_weights
from diffsg_amd.classifier_free_MSR import DDPM
name = sys.argv[1] if len(sys.argv) > 1 else "msr3"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0"); cfg = CONFIGS[name]; T, B = 20, 512
torch.manual_seed(0)
m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
d = DDPM(T, m, cfg["input_dim"], 10.0, 1.0 - generate_cosine_schedule(T), dev, (1, cfg["input_dim"]), None)
d.apply(init_weights); d.to(dev)
X = torch.rand(chunks * B, cfg["cond_dim"], device=dev)
d.sample(X[:B], 500.0); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    Y = torch.cat([d.sample(X[i:i + B], 500.0) for i in range(0, X.shape[0], B)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{name}: {chunks} chunks of {B} rows, T={T}: {dt*1e3:.2f} ms total, {dt/chunks*1e3:.3f} ms per chunk")
