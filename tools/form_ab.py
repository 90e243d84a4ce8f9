#!/usr/bin/env python3
"""Per-operator kernel time under two launch policies (the large-launch forms against the cooperative N-split form k_resblock_c at the
same size): python tools/form_ab.py [B] [iters]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from diffsg_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 6)
cond = torch.rand(B, 80, device=dev)
L = _lib.lib()
res = {}
for label, pol in (("large", (-1, -1)), ("coop", (1 << 20, -1)), ("large2", (-1, -1))):
    ddpm.model.set_launch_policy(*pol)
    ddpm.sample(cond, 1.0, seed=1)
    hd = ddpm.model.native_handle()
    names = []
    for i in range(L.dsg_op_count(hd)):
        nm = ctypes.create_string_buffer(64)
        L.dsg_op_info(hd, i, nm, None, None)
        names.append(nm.value.decode())
    lo, hi = ctypes.c_int(), ctypes.c_int()
    L.dsg_fused_range(hd, ctypes.byref(lo), ctypes.byref(hi))
    for i, name in enumerate(names):
        if lo.value < i < hi.value:
            continue
        ms = ctypes.c_float()
        for rep in range(2):
            _lib.check(L.dsg_time_op(hd, i, B, iters, ctypes.byref(ms), _lib.stream_ptr()))
        torch.cuda.synchronize()
        res.setdefault(name, {})[label] = ms.value * 1e3
print(f"B={B}: us per launch   large | coop (k_resblock_c, N-split, 4 waves per tile) | large again")
for n, r in res.items():
    print(f"{n:16s} {r.get('large', 0):8.1f} {r.get('coop', 0):8.1f} {r.get('large2', 0):8.1f}")
