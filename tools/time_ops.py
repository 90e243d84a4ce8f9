#!/usr/bin/env python3
"""Time several operators' kernels on realistic activations: python tools/time_ops.py [B] [iters] op1 op2 ..."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from diffsg_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
want = sys.argv[3:]
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 6)
cond = torch.rand(B, 80, device=dev)
ddpm.sample(cond, 1.0, seed=1)          # fills the workspace with real activations
L, hd = _lib.lib(), ddpm.model.native_handle()
names = []
for i in range(L.dsg_op_count(hd)):
    nm = ctypes.create_string_buffer(64)
    L.dsg_op_info(hd, i, nm, None, None)
    names.append(nm.value.decode())
lo, hi = ctypes.c_int(), ctypes.c_int()
L.dsg_fused_range(hd, ctypes.byref(lo), ctypes.byref(hi))
tot = 0.0
for i, name in enumerate(names):
    if want and name not in want:
        continue
    if not want and lo.value < i < hi.value:
        continue
    ms = ctypes.c_float()
    for rep in range(2):
        _lib.check(L.dsg_time_op(hd, i, B, iters, ctypes.byref(ms), _lib.stream_ptr()))
    torch.cuda.synchronize()
    tot += ms.value
    print(f"{name:16s} {ms.value*1e3:8.1f} us")
print(f"{'sum':16s} {tot*1e3:8.1f} us")

# the same operators launched ALTERNATELY, one launch per timing call: what a launch costs when another kernel ran just before it
# (instruction cache, L2 state) -- the situation inside a reverse step
if len(want) >= 2:
    acc = {n: 0.0 for n in want}
    rounds = 20
    for r in range(rounds + 2):
        for n in want:
            _lib.check(L.dsg_time_op(hd, names.index(n), B, 1, ctypes.byref(ms), _lib.stream_ptr()))
            if r >= 2:
                acc[n] += ms.value
    print("alternating, one launch per call:")
    for n in want:
        print(f"{n:16s} {acc[n] / rounds * 1e3:8.1f} us")
