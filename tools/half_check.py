import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, bench
dev = torch.device("cuda:0")
for B in (65536, 65536 + 32 * 5, 131072):
    for T in (1, 2, 5):
        ddpm = bench.build_model(dev, T)
        cond = torch.rand(B, 80, device=dev)
        outs = {}
        for val in (0, 1, 0, 1):
            ddpm.model.set_option("panel_half", val)
            y = ddpm.sample(cond, 1.0, seed=1)
            if val in outs:
                print(f"B={B} T={T} panel_half={val}: repeat identical: {torch.equal(outs[val], y)}")
            outs[val] = y
        d = (outs[0] - outs[1]).abs()
        print(f"B={B} T={T}: max rel diff {float(d.max() / outs[0].abs().max()):.3e}; rows differing: {int((d.max(dim=1).values > 0).sum())} of {B}", flush=True)
        if T == 1:
            bad = (d.max(dim=1).values > 1e-6 * outs[0].abs().max()).nonzero().flatten()
            print("   first bad rows:", bad[:20].tolist(), " tiles:", sorted(set((bad // 32).tolist()))[:20])
