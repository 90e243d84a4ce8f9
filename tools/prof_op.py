#!/usr/bin/env python3
"""Run one operator's kernel repeatedly on realistic activations (for rocprofv3 --pmc / --kernel-trace passes).
    python tools/prof_op.py [op_name] [B] [iters]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from diffsg_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "up.17.res"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 6)
if os.environ.get("PRECISION"):      # e.g. PRECISION=f32: the exact-float32 kernels
    ddpm.model.set_precision(os.environ["PRECISION"])
if os.environ.get("POLICY"):         # "coop_max_tiles,narrow_small_max_tiles" (dsg_set_launch_policy), e.g. POLICY=1048576,-1: the cooperative
    ddpm.model.set_launch_policy(*[int(v) for v in os.environ["POLICY"].split(",")])   # N-split form at every size
cond = torch.rand(B, 80, device=dev)
ddpm.sample(cond, 1.0, seed=1)          # fills the workspace with real activations
L, hd = _lib.lib(), ddpm.model.native_handle()
names = []
for i in range(L.dsg_op_count(hd)):
    nm = ctypes.create_string_buffer(64)
    L.dsg_op_info(hd, i, nm, None, None)
    names.append(nm.value.decode())
op = names.index(name)
ms = ctypes.c_float()
_lib.check(L.dsg_time_op(hd, op, B, iters, ctypes.byref(ms), _lib.stream_ptr()))
torch.cuda.synchronize()
print(f"{name}: {ms.value*1e3:.1f} us per launch at B={B}")
