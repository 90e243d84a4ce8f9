import ctypes, os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
ddpm = bench.build_model(dev, 20)
cond = torch.rand(B, 80, device=dev)
L = ctypes.CDLL(os.path.join(ROOT, "diffsg_amd", "libdiffsg_hip.so"))
buf = (ctypes.c_ulonglong * 8192)()
for it in range(3):
    ddpm.sample(cond, 1.0, seed=1); torch.cuda.synchronize()
    n = L.dsg_dbg_fetch(buf, 8192)
st = [(buf[i] & 0xff, buf[i] >> 8) for i in range(n)]
ops = []
for tag, t in st:
    if tag == 1: ops.append([])
    if ops: ops[-1].append((tag, t))
ops = ops[-17:]      # the last step's narrow run
for k, op in enumerate(ops):
    print(f"op {k:2d} total {op[-1][1]-op[0][1]:6d}  " + " ".join(f"{tag}:{t - p:5d}" for (tag, t), (_, p) in zip(op[1:], op[:-1])))
print("gaps:", [ops[i + 1][0][1] - ops[i][-1][1] for i in range(len(ops) - 1)])
