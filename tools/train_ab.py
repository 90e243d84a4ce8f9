#!/usr/bin/env python3
"""Same-box A-B of the host-side pieces of the training step at 32 768 rows (MSR-80c): python tools/train_ab.py [rows] [steps]
  native Adam (dsg_adam_step) vs torch's fused kernel; the three draws on a side stream vs on the caller's stream; device-side draws.
Also prints the host time of one step enqueued into an empty queue (how far the host runs ahead of the GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
from diffsg_amd.train import FlatAdam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 20)
opt = FlatAdam(ddpm, lr=0.005)
cond = torch.rand(B, 80, device=dev); y = torch.rand(B, 80, device=dev) * 0.25
def one():
    loss = ddpm(y, cond); loss.backward(); opt.step(); opt.zero_grad(); return loss
def run(tag):
    for _ in range(5): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize(); h0 = time.perf_counter(); one(); h1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{tag:58s} {dt*1e3:.3f} ms/step   host enqueue of one step {1e3*(h1-h0):.3f} ms", flush=True)
if os.environ.get("SPLIT_AB"):
    # split of the batch over two handles / streams (DDPM.train_split_min_rows), with the settings bench.py's train leg uses
    FlatAdam.native_step = True; ddpm.device_draws = 1
    for rnd in range(3):
        for sp in (None, 16384):
            type(ddpm).train_split_min_rows = sp
            run(f"round {rnd}: B={B} train_split_min_rows={sp}")
    sys.exit(0)
if os.environ.get("PART_AB"):
    # the narrow run's weight gradients as a third early part beside the end of the chain (DSG_OPT_WGRAD_NARROW_PART)
    FlatAdam.native_step = True; ddpm.device_draws = 1
    for rnd in range(2):
        for v in (0, 1):
            ddpm.model.set_option("wgrad_narrow_part", v)
            run(f"round {rnd}: B={B} wgrad_narrow_part={v}")
    sys.exit(0)
if os.environ.get("TIME_AB"):
    # the time-path backward beside the last weight-gradient launch (DSG_OPT_TRAIN_TIME_BESIDE) against behind it
    FlatAdam.native_step = True; ddpm.device_draws = 1
    for rnd in range(3):
        for v in (0, 1):
            ddpm.model.set_option("train_time_beside", v)
            run(f"round {rnd}: B={B} train_time_beside={v}")
    sys.exit(0)
for rnd in range(2):
    for nat in (False, True):
        for side in (False, True):
            FlatAdam.native_step = nat; ddpm.draws_on_side_stream = side; ddpm.device_draws = None
            run(f"round {rnd}: native_adam={nat} draws_on_side_stream={side}")
    FlatAdam.native_step = True; ddpm.device_draws = 1
    run(f"round {rnd}: native_adam=True device_draws")
