#!/bin/bash
# Same-box A-B-A-B-A-B of two source trees on the latency-bound shapes: the tree itself and a copy of an older one under ab_old/ (git archive <rev> |
# tar -x -C ab_old, built there: ab_old/ is git-ignored but travels with gpurun).  Per round and tree: the T = 20 sample() call of MSR-3c 8 192 rows
# (BASELINE config 2) and of MSR-80c 512 rows (tools/small_batch.py), and 40 training steps at 32 768 rows.   usage (GPU box): bash tools/ab_small.sh
for r in 1 2 3; do
  for tree in ab_old .; do
    (cd $tree && echo "== $tree" && python3 tools/small_batch.py msr3 8192 2>&1 | grep "ms per" && python3 tools/small_batch.py msr80 512 2>&1 | grep "ms per" && PART_AB= python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import torch, bench
from diffsg_amd.train import FlatAdam
B=32768
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 20)
opt = FlatAdam(ddpm, lr=0.005)
FlatAdam.native_step = True; ddpm.device_draws = 1
cond = torch.rand(B, 80, device=dev); y = torch.rand(B, 80, device=dev) * 0.25
def one():
    loss = ddpm(y, cond); loss.backward(); opt.step(); opt.zero_grad(); return loss
for _ in range(6): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): one()
torch.cuda.synchronize(); print(f"train B={B}: {(time.perf_counter()-t0)/40*1e3:.3f} ms/step")
PY
    )
  done
done
