#!/usr/bin/env python3
"""N MSR-80c training steps at 32768 rows (for rocprofv3 kernel-trace): python tools/train_prof.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 20)
from diffsg_amd.train import FlatAdam
opt = FlatAdam(ddpm, lr=0.005) if not os.environ.get('PLAIN_ADAM') else torch.optim.Adam(ddpm.parameters(), lr=0.005, fused=True)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
cond = torch.rand(B, 80, device=dev); y = torch.rand(B, 80, device=dev) * 0.25
if os.environ.get('POLICY'):          # "coop_max_tiles,narrow_small_max_tiles" (dsg_set_launch_policy), e.g. POLICY=1024,1024
    ddpm.model.set_launch_policy(*[int(v) for v in os.environ['POLICY'].split(',')])
if os.environ.get('DEVICE_DRAWS'):     # ts / noise / mask drawn inside the library (dsg_train_step_seeded)
    ddpm.device_draws = 1
def one():
    loss = ddpm(y, cond); loss.backward(); opt.step(); opt.zero_grad(); return loss
for _ in range(3): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): one()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"train: {dt/steps*1e3:.3f} ms/step, {B*steps/dt/1e6:.2f} M samples/s")
if os.environ.get('PHASES'):
    import ctypes
    from diffsg_amd import _lib
    L, hd = _lib.lib(), ddpm.model.native_handle()
    _lib.check(L.dsg_train_profile_enable(hd, 1))
    acc = [0.0] * 5
    for _ in range(5):
        one(); ms5 = (ctypes.c_float * 5)(); _lib.check(L.dsg_train_profile(hd, ms5)); acc = [a + float(v) for a, v in zip(acc, ms5)]
    print("phases ms (fwd, act-bwd, colsum, wgrad, tail):", [round(a / 5, 3) for a in acc])
