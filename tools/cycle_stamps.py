#!/usr/bin/env python3
"""Phase times inside the hot kernels from cycle stamps (library built with -DDSG_CYCLE_STAMPS: tools/cycle_stamps.sh does build, run, rebuild).
   python tools/cycle_stamps.py [sample_rows] [train_rows]
Tags  0x11-0x19 narrow forward block (tile 0 of the fused run): start, after LN1 statistics + requests, after stage 1, after un-scale + time bias,
                after stage 2, after c2 + condition term, after stage 3, after shortcut / residual, after the output statistics
      0x21-0x29 narrow backward block: start, dL/d(out) + column sums, W3^T GEMM, LN3 backward + dh2, W2^T GEMM, LN2 backward + dh1, W1^T GEMM,
                LN1 backward, shortcut
      0x31-0x3b k_wide128_h (linear-shortcut block, workgroup 0, wave 0): start, vectors staged, statistics + program, table, barrier, stage 1,
                stage 2, condition term, stage 3, shortcut, statistics + store"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from diffsg_amd import _lib
Bs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
_lib.lib()
L = ctypes.CDLL(os.path.join(ROOT, "diffsg_amd", "libdiffsg_hip.so"))
if not hasattr(L, "dsg_stamps_fetch"):
    raise SystemExit("library built without -DDSG_CYCLE_STAMPS: run tools/cycle_stamps.sh")
buf = (ctypes.c_ulonglong * 8192)()

def fetch():
    n = L.dsg_stamps_fetch(buf, 8192)
    return sorted(((buf[i] >> 16, buf[i] & 0xffff) for i in range(n)))

def table(stamps, lo, first, name, keep):
    runs = []
    for t, tag in stamps:
        if not (lo <= tag < lo + 16): continue
        if tag == first: runs.append([])
        if runs: runs[-1].append((tag, t))
    runs = [r for r in runs if len(r) > 2][-keep:]
    for k, r in enumerate(runs):
        print(f"{name} {k:2d}: total {r[-1][1] - r[0][1]:7d}  " + " ".join(f"{tag:x}:{t - p:6d}" for (tag, t), (_, p) in zip(r[1:], r[:-1])))

ddpm = bench.build_model(dev, 20)
cond = torch.rand(Bs, 80, device=dev)
for _ in range(2):
    ddpm.sample(cond, 1.0, seed=1); torch.cuda.synchronize(); fetch()
ddpm.sample(cond, 1.0, seed=1); torch.cuda.synchronize()
st = fetch()
print(f"== sampling, {Bs} rows, last reverse step(s)")
table(st, 0x30, 0x31, "wide up-128 block", 3)
table(st, 0x10, 0x11, "narrow fwd block", 17)
cond = torch.rand(Bt, 80, device=dev); y = torch.rand(Bt, 80, device=dev) * 0.25
for _ in range(2):
    loss = ddpm(y, cond); loss.backward(); torch.cuda.synchronize(); fetch()
loss = ddpm(y, cond); loss.backward(); torch.cuda.synchronize()
st = fetch()
print(f"== training step, {Bt} rows")
table(st, 0x10, 0x11, "narrow fwd block", 17)
table(st, 0x20, 0x21, "narrow bwd block", 17)
