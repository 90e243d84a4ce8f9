#!/usr/bin/env python3
"""Where a reverse step of a SMALL batch spends its time inside k_unet_tile (library built with -DDSG_CYCLE_STAMPS): cycles per operator of
workgroup 0 (its first wave), last step of a call.   DSG_EXTRA_CXXFLAGS=-DDSG_CYCLE_STAMPS python tools/tile_stamps.py [config] [rows]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from weights import CONFIGS
from diffsg_amd import UNet1D, generate_cosine_schedule, init_weights, _lib
from diffsg_amd.classifier_free_MSR import DDPM
name = sys.argv[1] if len(sys.argv) > 1 else "msr3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = 8
dev = torch.device("cuda:0")
cfg = CONFIGS[name]
torch.manual_seed(0)
m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
D = cfg["input_dim"]
d = DDPM(T, m, D, 10.0, 1.0 - generate_cosine_schedule(T), dev, (1, D), None)
d.apply(init_weights); d.to(dev)
_lib.lib()
L = ctypes.CDLL(os.path.join(ROOT, "diffsg_amd", "libdiffsg_hip.so"))
if not hasattr(L, "dsg_stamps_fetch"):
    raise SystemExit("library built without -DDSG_CYCLE_STAMPS")
buf = (ctypes.c_ulonglong * 8192)()
def fetch():
    n = L.dsg_stamps_fetch(buf, 8192)
    return sorted(((buf[i] >> 16, buf[i] & 0xffff) for i in range(n)))
cond = torch.rand(B, cfg["cond_dim"], device=dev)
d.sample(cond, 1.0, seed=1); torch.cuda.synchronize(); fetch()
d.sample(cond, 1.0, seed=2, use_graph=False); torch.cuda.synchronize()
allst = fetch()
st = [(t, tag) for t, tag in allst if 0x1000 <= tag < 0x2000]
hd = d.model.native_handle()
Lb = _lib.lib()
names = []
for i in range(Lb.dsg_op_count(hd)):
    nm = ctypes.create_string_buffer(64); Lb.dsg_op_info(hd, i, nm, None, None); names.append(nm.value.decode())
# the last launch: from its first operator stamp to 0x1fff
ends = [k for k, (t, tag) in enumerate(st) if tag == 0x1fff]
if not ends: raise SystemExit("no k_unet_tile stamps (tile_step off, or the batch is above coop_max_tiles)")
hi = ends[-1]; lo = ends[-2] + 1 if len(ends) > 1 else 0
run = st[lo:hi + 1]
tot = run[-1][0] - run[0][0]
print(f"{name} B={B}: one k_unet_tile launch (workgroup 0) = {tot} cycles")
for (t0, tag), (t1, _) in zip(run[:-1], run[1:]):
    i = tag - 0x1000
    print(f"  {names[i] if i < len(names) else '?':14s} {t1 - t0:8d} cycles  {100.0 * (t1 - t0) / tot:5.1f} %")

# phases inside the cooperative wide blocks (tags 0x2001-0x200d of tile 0, slice 0), last launch
ph = [(t, tag) for t, tag in allst if 0x2000 <= tag < 0x2100 and run[0][0] <= t <= run[-1][0]]
names2 = {0x2001: "LN1 stats", 0x2002: "loads + transform 1", 0x2003: "barrier", 0x2004: "stage-1 MFMAs", 0x2005: "epilogue 1 + stats + barrier",
          0x2006: "transform 2", 0x2007: "barrier", 0x2008: "stage-2 MFMAs + epilogue", 0x2009: "cond + stats + barrier", 0x200a: "transform 3",
          0x200b: "barrier", 0x200c: "stage 3 (+ shortcut)", 0x200d: "stats + 2 barriers"}
blocks = []
for t, tag in ph:
    if tag == 0x2001: blocks.append([])
    if blocks: blocks[-1].append((t, tag))
for k, b in enumerate(blocks):
    print(f"  wide block {k}: " + "  ".join(f"{names2[tag]}: {t - p}" for (t, tag), (p, _) in zip(b[1:], b[:-1])))
