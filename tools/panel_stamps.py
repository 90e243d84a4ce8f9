#!/usr/bin/env python3
"""Phase stamps of the persistent panel kernel (library built with -DDSG_CYCLE_STAMPS; run with DSG_EXTRA_CXXFLAGS=-DDSG_CYCLE_STAMPS):
    python tools/panel_stamps.py [op] [B]
Tags (stage S: 1 stage 1, 2 stage 2, 3 stage 3, 4 shortcut): 01 group start; S0 = the stage's row statistics done; S1 = step 0's operand
prepared (outside the MFMA stream); S4 = panel begun (weights waited, barrier, next panel issued); S2 = panel's MFMA stream done (with the
next step's operand prepared inside it); 13 / 23 / 43 = stage epilogue (un-scale, condition term); 50 = stored.  The stamps write LDS:
the stamped binary's RESULTS are not valid, only its timing."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from diffsg_amd import _lib
op_name = sys.argv[1] if len(sys.argv) > 1 else "up.17.res"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 6)
cond = torch.rand(B, 80, device=dev)
ddpm.sample(cond, 1.0, seed=1)
L, hd = _lib.lib(), ddpm.model.native_handle()
names = []
for i in range(L.dsg_op_count(hd)):
    nm = ctypes.create_string_buffer(64)
    L.dsg_op_info(hd, i, nm, None, None)
    names.append(nm.value.decode())
ms = ctypes.c_float()
buf = (ctypes.c_ulonglong * 8192)()
L2 = ctypes.CDLL(os.path.join(ROOT, "diffsg_amd", "libdiffsg_hip.so"))
_lib.check(L.dsg_time_op(hd, names.index(op_name), B, 5, ctypes.byref(ms), _lib.stream_ptr()))
torch.cuda.synchronize()
n = L2.dsg_stamps_fetch(buf, 8192)
print(f"{op_name}: {ms.value*1e3:.1f} us per launch; {n} stamp slots")
if n >= 8 * 128 + 4:
    t0, r0, t1, r1 = (buf[8 * 128 + k] for k in range(4))
    if r1 > r0:
        print(f"clock under this kernel (workgroup 0, whole launch): {t1 - t0} shader cycles in {(r1 - r0) * 10} ns = {(t1 - t0) / ((r1 - r0) * 10.0):.3f} GHz")
for w in range(8):
    st = [(buf[w * 128 + k] >> 16, buf[w * 128 + k] & 0xffff) for k in range(128) if buf[w * 128 + k]]
    if not st:
        continue
    t0 = st[0][0]
    print(f"wave {w}: " + " ".join(f"{tag:02x}:{t - p}" for (t, tag), (p, _) in zip(st[1:], st[:-1])) + f"  | total {st[-1][0] - t0}")

# merged absolute timeline of the two waves that share SIMD 0 (waves 0 and 4), second tile group
ev = []
for w in (0, 4):
    for k in range(128):
        v = buf[w * 128 + k]
        if v:
            ev.append((v >> 16, w, v & 0xffff))
ev.sort()
t0 = ev[0][0]
half = [e for e in ev if e[0] - t0 > 0][: 2 * 40]
print("timeline (cycle, wave, tag reached):")
print(" ".join(f"{t - t0}:w{w}:{tag:02x}" for t, w, tag in ev[:110]))
