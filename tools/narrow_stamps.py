#!/usr/bin/env python3
"""Per-operator cycle stamps of tile 0 inside k_fused_narrow_lds (measurement build, -DDSG_CYCLE_STAMPS: tools/narrow_stamps.sh):
   tags 0x400 kernel entry (wave 0 of workgroup 0), 0x401 image staged, 0x410 + i operator i of the run starts, 0x480 the float32 section starts,
   0x4ff the wave is done.   python tools/narrow_stamps.py [rows]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from diffsg_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
_lib.lib()
L = ctypes.CDLL(os.path.join(ROOT, "diffsg_amd", "libdiffsg_hip.so"))
buf = (ctypes.c_ulonglong * 8192)()
def fetch():
    n = L.dsg_stamps_fetch(buf, 8192)
    return sorted(((buf[i] >> 16, buf[i] & 0xffff) for i in range(n)))
ddpm = bench.build_model(dev, 6)
cond = torch.rand(B, 80, device=dev)
for _ in range(2):
    ddpm.sample(cond, 1.0, seed=1); torch.cuda.synchronize(); fetch()
ddpm.sample(cond, 1.0, seed=1); torch.cuda.synchronize()
st = [(t, tag) for t, tag in fetch() if 0x400 <= tag <= 0x4ff]
# launches: split at every 0x400
runs = []
for t, tag in st:
    if tag == 0x400: runs.append([])
    if runs: runs[-1].append((tag, t))
for r in runs[-4:]:
    t0 = r[0][1]
    print(f"launch: total {r[-1][1] - t0} cycles: " + " ".join(f"{tag:x}:+{t - p}" for (tag, t), (_, p) in zip(r[1:], r[:-1])))
