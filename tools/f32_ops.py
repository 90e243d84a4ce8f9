#!/usr/bin/env python3
"""Per-operator kernel time of the exact-float32 path: python tools/f32_ops.py [B] [iters]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from diffsg_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 6)
ddpm.model.set_precision("f32")
cond = torch.rand(B, 80, device=dev)
ddpm.sample(cond, 1.0, seed=1)
L, hd = _lib.lib(), ddpm.model.native_handle()
tot = 0.0
for i in range(L.dsg_op_count(hd)):
    nm = ctypes.create_string_buffer(64)
    L.dsg_op_info(hd, i, nm, None, None)
    ms = ctypes.c_float()
    for rep in range(2):
        _lib.check(L.dsg_time_op(hd, i, B, iters, ctypes.byref(ms), _lib.stream_ptr()))
    torch.cuda.synchronize()
    tot += ms.value
    print(f"{nm.value.decode():16s} {ms.value*1e3:8.1f} us")
print(f"{'sum':16s} {tot*1e3:8.1f} us")
