#!/bin/bash
# Same-box A-B-A-B of two source trees (the tree itself and a copy of an older one under ab_old/, each with its own built library):
#   bash tools/tree_ab.sh [rows] [T] [rounds]     -> steps/s of `tools/option_ab.py`-style timing per tree and round
ROWS=${1:-65536}; T=${2:-50}; N=${3:-2}
for r in $(seq 1 $N); do
  for tree in ab_old .; do
    (cd $tree && python3 - "$ROWS" "$T" "$tree" <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import torch, bench
B, T, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, T)
cond = torch.rand(B, 80, device=dev)
for _ in range(3): ddpm.sample(cond, 1.0, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): y = ddpm.sample(cond, 1.0, seed=1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4 / T
ddpm.sample(cond, 1.0, seed=1, profile=True); torch.cuda.synchronize()
ops = {r[0][:14]: round(r[3] / T * 1e3, 1) for r in ddpm.op_profile() if r[3] / T > 8e-3}
print(f"{tag:7s} {dt*1e3:.4f} ms/step = {1/dt:.1f} steps/s  {ops}", flush=True)
PY
    )
  done
done
