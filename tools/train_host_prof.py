#!/usr/bin/env python3
"""Host-side profile of the training step at a small batch (where the step is launch/host bound):
python tools/train_host_prof.py [rows] [steps]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from diffsg_amd.train import FlatAdam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 20)
opt = FlatAdam(ddpm, lr=0.005)
cond = torch.rand(B, 80, device=dev); y = torch.rand(B, 80, device=dev) * 0.25
def one():
    loss = ddpm(y, cond); loss.backward(); opt.step(); opt.zero_grad(); return loss
for _ in range(10): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): one()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"train B={B}: {dt/steps*1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): one()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
