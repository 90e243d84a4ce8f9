#!/usr/bin/env python3
"""bench.py -- DDPM reverse-sampling throughput on MSR-80c (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one reverse step of DDPM.sample (classifier_free_MSR.py:124-137) over one synthetic batch:
2 denoiser forwards (unconditional + conditional) + CFG combine + ancestral update (+ the global renorm on the first
4 steps), B = 65 536 rows x D = C = 80, float32, weights = `init_weights` state (seed 0), cond ~ U[0,1), noise from the
device Philox stream; inputs are resident in HBM before the timed region.  Reverse sampling shards by rows with no
collective: every rank runs its own B-row batch (weak scaling) and `value` counts the B-row steps the
whole job finished per second.

Arithmetic: float32 in HBM and in every accumulator; the GEMMs run as hi/lo fp16 splits (22 significant bits per operand)
on v_mfma_f32_32x32x16_f16 (`diffsg_amd/csrc/dsg_split.hpp`), which measures as accurate against the reference as the
exact v_mfma_f32_32x32x2_f32 path (1.3e-6 vs 1.4e-6 worst relative error on the parity cases) and is what `dtype`
"f32(2xf16-split MFMA, f32 accumulate)" names.  `--precision f32` times the exact-f32 MFMA path instead.

Printed JSON (rank 0): the driver contract keys plus
  roofline      dominant kernel (the proj_dim-wide up blocks): mean launch time from HIP events recorded around each launch
                on the launch stream in a second, eager run of the same K steps.  `achieved` = EXECUTED matrix-core FLOP per
                second: the split path issues 3 v_mfma_f32_32x32x16_f16 per float32 product, so 3 x the algorithmic FLOP
                (SURVEY 8(d) per-row figure x rows per launch) / launch time, against `peak` = 2500 TFLOP/s, the dense f16
                MFMA rate of the unit the kernel runs on (`--precision f32`: 1 x, against the 157.3 TFLOP/s f32 MFMA rate).
                `algorithmic_tflops` / `frac_f32_equiv` keep the float32-equivalent figure (algorithmic FLOP only, against
                the f32 roof SURVEY 8(d) names) under their own keys.  `traffic` (HBM bytes per launch) and `valu_per_mfma`
                are NOT measured in this run: they are imported from the committed rocprofv3 PMC summary named in
                `traffic_source` (separate --pmc passes, FETCH_SIZE doubled per the gfx950 correction).
                `step_roofline` gives the whole step against the same roofs.
  cpu_baseline  the CPU oracle (oracle/ddpm_oracle.py, the bit-checked restatement of the reference) timed on this
                box's host cores on a bounded sample, converted to the same unit.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters (fp32 vector = fp32 MFMA)
PEAK_F16_TFLOPS = 2500.0  # dense f16/bf16 MFMA (the 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0     # spec; 6290 measured
MSR80 = dict(input_dim=80, proj_dim=128, cond_dim=80, dims=(64, 32, 16, 8), n_blocks=2)
# Algorithmic work per sample-step: 2 passes x trunk MACs.  The time path (depends only on the step) and the condition
# embeddings (depend only on the row, not on the step) are computed once per sample() call, outside the step; SURVEY
# 8(d) still counted the conditional pass's cond GEMMs (+100 480 MAC) because it hoisted only the time path.
F_ALG = 2 * (2 * 566_400)             # FLOP / row / step
F_ALG_SURVEY = 2 * (2 * 566_400 + 100_480)   # SURVEY 8(d)'s figure (the conditional pass's cond GEMMs counted): prices the 973-steps/s fp32 roof
BYT_ALG = 4 * (3 * 80 + 80)           # B / row / step


def build_model(device, T, seed=0):
    from diffsg_amd import UNet1D, generate_cosine_schedule, init_weights
    from diffsg_amd.classifier_free_MSR import DDPM
    torch.manual_seed(seed)
    model = UNet1D(**MSR80, is_attn=(False,) * 4)
    ddpm = DDPM(T, model, 80, 20.0, 1.0 - generate_cosine_schedule(T), device, (1, 80), None, 0.1, 0.9999, 10, 5, False)
    ddpm.apply(init_weights)
    return ddpm.to(device)


def self_launch(a):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as CHILD processes -- before this process touches the
    GPU in any way -- with the env torch.distributed.run would give them, relay rank 0's JSON line, exit with the worst code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, abs(p.wait()))
    sys.stdout.write(out0)
    sys.stdout.flush()
    raise SystemExit(rc)


def box_calibrate():
    """dsg_box_calibrate on the current device: {mfma_tflops, mix_gslots, copy_gbs} of three fixed probes (how to compare
    two lines measured on different boxes)."""
    import ctypes
    from diffsg_amd import _lib
    o = (ctypes.c_float * 4)()
    try:
        _lib.check(_lib.lib().dsg_box_calibrate(o, _lib.stream_ptr()))
    except RuntimeError as e:            # the probe allocates 512 MiB beside the batch: a failed probe must not abort the benchmark
        return {"error": str(e)[:200]}
    return {"mfma_tflops": round(o[0], 1), "mix_gslots": round(o[1], 2), "copy_gbs": round(o[2], 1), "panel_gslots": round(o[3], 2)}


def cpu_baseline(sample_rows=4096, steps=2, B_ref=65536):
    """Oracle timed on the host: `steps` reverse steps of a `sample_rows`-row batch.  The thread count is the best
    of {32, 16, 8} on a one-step probe (hundreds of small ATen ops oversubscribe a 128-core host: all cores is slower)."""
    ncpu = os.cpu_count() or 1
    best = None
    for nt in sorted({min(ncpu, 32), min(ncpu, 16), min(ncpu, 8)}, reverse=True):
        r = _cpu_baseline_run(nt, sample_rows, 1, B_ref)
        if best is None or r["row_steps_per_s"] > best["row_steps_per_s"]:
            best = r
    return _cpu_baseline_run(best["cores"], sample_rows, steps, B_ref)


def _cpu_baseline_run(nthreads, sample_rows, steps, B_ref):
    torch.set_num_threads(nthreads)
    from oracle import ddpm_oracle as O
    from weights import synth_weights
    T = 20
    plan = O.unet_plan(MSR80["input_dim"], MSR80["proj_dim"], MSR80["cond_dim"], MSR80["dims"], MSR80["n_blocks"])
    p = {k: torch.from_numpy(v) for k, v in synth_weights(O.state_shapes(plan), 0, "init").items()}
    bufs = O.schedule_buffers(1.0 - O.cosine_betas(T))
    g = torch.Generator().manual_seed(0)
    cond = torch.rand(sample_rows, 80, generator=g)
    y = torch.randn(sample_rows, 80, generator=g)
    z = torch.randn(sample_rows, 80, generator=g)
    with torch.no_grad():
        O.sample_step(p, plan, bufs, T, 10, y, cond, 1.0, z)  # warm-up
        t0 = time.perf_counter()
        for i in range(steps):
            y, _ = O.sample_step(p, plan, bufs, T, 10 - i, y, cond, 1.0, z)
        dt = time.perf_counter() - t0
    row_steps = sample_rows * steps / dt
    return {"value": row_steps / B_ref, "unit": "steps/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{steps} reverse steps of a {sample_rows}x80 batch (T=20, omega=1) with oracle/ddpm_oracle.py on "
                      f"{torch.get_num_threads()} torch threads = {row_steps:.0f} row-steps/s, scaled to B={B_ref}",
            "row_steps_per_s": row_steps}


def config2_leg(dev):
    """BASELINE.json config 2: MSR 3-cell reverse sampling, T = 1000, batch 8192, one GPU -- one timed DDPM.sample call after one
    untimed call (tables, graphs), same arithmetic as the headline."""
    from weights import CONFIGS
    from diffsg_amd import UNet1D, generate_cosine_schedule, init_weights
    from diffsg_amd.classifier_free_MSR import DDPM
    cfg, T, B = CONFIGS["msr3"], 1000, 8192
    torch.manual_seed(0)
    m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
    d = DDPM(T, m, cfg["input_dim"], 10.0, 1.0 - generate_cosine_schedule(T), dev, (1, cfg["input_dim"]), None)
    d.apply(init_weights)
    d.to(dev)
    cond = torch.rand(B, cfg["cond_dim"], device=dev)
    d.sample(cond, 1.0, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = d.sample(cond, 1.0, seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "MSR-3c CFG reverse sampling, batch 8192 x D=C=3, T=1000, omega=1 (BASELINE config 2)", "value": T / dt, "unit": "steps/s",
            "ms_per_step": dt / T * 1e3, "row_steps_per_s": B * T / dt, "finite": bool(torch.isfinite(y).all())}


# trunk / cond MACs per sample and forward.  Sampling work per row-step as the headline counts it (both passes' trunks; the
# condition embeddings are computed once per call, outside the step); training 3 x 2 x (trunk + cond) with the time path tabulated over T rows.
OTHER_CONFIGS = {
    "co3": dict(trunk=329_024, cond=11_736, lr=0.005, label="CO-3n (classifier_free_CO.py:218-219: proj 64, dims (64,32,16,8), n_blocks 3; D=3, C=9)"),
    "nu3": dict(trunk=60_864, cond=2_736, lr=0.004, label="NU-3u (classifier_free_NU.py:230-231: proj 32, dims (32,16,8), n_blocks 2; D=5, C=6)"),
}


def _build_other(name, dev, T):
    from weights import CONFIGS
    from diffsg_amd import UNet1D, generate_cosine_schedule, init_weights
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
    D = cfg["input_dim"]
    alphas = 1.0 - generate_cosine_schedule(T)
    if name == "co3":
        from diffsg_amd.classifier_free_CO import DDPM
        d = DDPM(T, m, D, alphas, dev, (1, D), None)
    else:
        from diffsg_amd.classifier_free_NU import DDPM
        d = DDPM(T, m, D - 2, 18.0, alphas, dev, (1, D), None)
    d.apply(init_weights)
    return d.to(dev), cfg


def configs_leg(dev, split=True):
    """BASELINE.json configs 3 and 4 (CO 3-node, NOMA-UAV 3-user: "train + sample, 1 x MI355X"): reverse sampling at 8 192 rows, T = 200,
    omega = 1 (one timed call behind one untimed call) and training at a local batch of 32 768 rows and at the shipped 512
    (classifier_free_CO.py:236-248, classifier_free_NU.py:248-260: forward + backward + Adam + re-pack; device-side draws) -- each with its
    algorithmic-FLOP fraction of the fp32 roof SURVEY 8(d) prices the path against.  Bounded: ~1 s of timed work for both configs."""
    from diffsg_amd.train import FlatAdam
    out = {}
    for name, spec in OTHER_CONFIGS.items():
        T, B = 200, 8192
        d, cfg = _build_other(name, dev, T)
        cond = torch.rand(B, cfg["cond_dim"], device=dev)
        d.sample(cond, 1.0, seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = d.sample(cond, 1.0, seed=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        f_step = 2 * 2 * spec["trunk"]
        rec = {"workload": spec["label"],
               "sample": {"rows": B, "T": T, "steps_per_s": T / dt, "ms_per_step": dt / T * 1e3, "row_steps_per_s": B * T / dt,
                          "f_alg_per_row_step": f_step, "algorithmic_tflops": f_step * B * T / dt / 1e12,
                          "frac_fp32_roof": f_step * B * T / dt / 1e12 / PEAK_F32_TFLOPS,
                          "frac_mfma_unit": (3.0 if split else 1.0) * f_step * B * T / dt / 1e12 / (PEAK_F16_TFLOPS if split else PEAK_F32_TFLOPS),
                          "finite": bool(torch.isfinite(y).all())},
               "train": {}}
        del d
        f_train = 3 * 2 * (spec["trunk"] + spec["cond"])
        for Bt, n in ((32768, 12), (512, 40)):
            dt_, cfg = _build_other(name, dev, 20)
            opt = FlatAdam(dt_, lr=spec["lr"])
            dt_.device_draws = 77
            D = cfg["input_dim"]
            c = torch.rand(Bt, cfg["cond_dim"], device=dev)
            yv = torch.rand(Bt, D, device=dev)

            def one():
                loss = dt_(yv, c)
                loss.backward()
                opt.step()
                opt.zero_grad()
                return loss
            for _ in range(4):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                loss = one()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            rec["train"][f"batch_{Bt}"] = {"samples_per_s": Bt * n / el, "ms_per_step": el / n * 1e3, "steps": n, "T": 20,
                                           "f_alg_per_sample": f_train, "algorithmic_tflops": Bt * n / el * f_train / 1e12,
                                           "frac_fp32_roof": Bt * n / el * f_train / 1e12 / PEAK_F32_TFLOPS,
                                           "final_loss": float(loss.detach())}
            del dt_, opt
        out[name] = rec
    return out


def train_leg(dev, dist, rank, world, B, steps, barrier, warmup=8, use_graph=True):
    """MSR-80c training throughput (BASELINE config 5 shape): per GPU `B` rows, T=20, Adam(lr 5e-3); one step =
    DDPM.forward (q_sample + denoiser forward + backward in libdiffsg_hip) + ONE all-reduce of the flat 6.6 MB gradient
    bucket (RCCL, world > 1) + Adam.step + re-pack of the updated weights."""
    ddpm = build_model(dev, 20)
    from diffsg_amd.train import FlatAdam
    opt = FlatAdam(ddpm, lr=0.005)
    # every rank draws its own ts / noise / mask: device-side Philox draws inside the fused step, the rank folded into the key
    # (DDPM.device_draws; distributions checked against MSR.py:101-107 by tests/test_gpu_parity.py).  build_model() seeds torch's
    # generator identically in every process: draws from it would repeat across ranks.
    ddpm.device_draws = 1000 + rank
    g = torch.Generator().manual_seed(100 + rank)
    cond = torch.rand(B, 80, generator=g).to(dev)
    y = (torch.rand(B, 80, generator=g) * (20.0 / 80)).to(dev)

    def one():
        loss = ddpm(y, cond)
        loss.backward()
        ddpm.allreduce_grads()
        opt.step()
        opt.zero_grad()
        return loss
    for _ in range(warmup):
        one()
    # one GPU: the whole step (draws, forward + backward, Adam, zero_grad, re-pack) replayed as ONE captured graph (train.StepGraph: the same
    # kernels on the same operands, bit-identical to the eager loop); data parallel: the eager loop with its one all-reduce per step
    sg, graph_note = None, None
    if use_graph and world == 1:
        from diffsg_amd.train import StepGraph
        try:
            sg = StepGraph(ddpm, opt, y, cond, warmup=2)
            for _ in range(3):
                sg.step()
        except Exception as e:            # a capture the runtime refuses must not cost the benchmark its training leg: time the eager loop
            graph_note = f"step graph unavailable ({type(e).__name__}: {str(e)[:120]}): eager launches"
            if sg is not None:
                sg.close()
            sg = None
            ddpm._call_dev = None
            opt._dyn = None
            torch.cuda.synchronize()
    step_fn = sg.step if sg is not None else one
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step_fn()
    t_host = time.perf_counter() - t0          # the host is done enqueueing here; the barrier below waits for the device
    barrier()
    dt = time.perf_counter() - t0
    if sg is not None:
        sg.close()
    from diffsg_amd import parallel as par
    ev = par.run_evidence(dev, dt, steps)            # ranks_seen, per-rank rates; the timed figure is the slowest rank's
    dt = ev["max_seconds"]
    # the bucket every rank holds after the step's all-reduce must be the same bits everywhere
    ddpm(y, cond).backward()
    ddpm.allreduce_grads()
    eq, csum = par.bucket_checksum_equal(ddpm.grad_bucket)
    opt.zero_grad()
    # algorithmic FLOP per sample: 3 x 2 x (trunk + cond) MACs with the time path tabulated over T rows
    f_train = 3 * 2 * (566_400 + 100_480)
    sps = world * B * steps / dt
    roof = None
    # phase times of the fused step from HIP events on the launch stream (dsg_train_profile): a few extra untimed steps, on
    # every rank (the step holds a collective), read on rank 0
    import ctypes
    from diffsg_amd import _lib
    L, hd = _lib.lib(), ddpm.model.native_handle()
    _lib.check(L.dsg_train_profile_enable(hd, 1))
    acc = [0.0] * 5
    n_prof = 5
    for _ in range(n_prof):
        one()
        ms5 = (ctypes.c_float * 5)()
        _lib.check(L.dsg_train_profile(hd, ms5))
        acc = [x + float(v) for x, v in zip(acc, ms5)]
    _lib.check(L.dsg_train_profile_enable(hd, 0))
    if rank == 0:
        fwd, bwd, cs, wg, tail = [x / n_prof for x in acc]
        # dominant kernel of the step: the grouped weight-gradient launch k_wgrad_h, dW = G^T A for every Linear:
        # 2 x (trunk + cond + T-row time table) MAC-FLOP per row, 3 f16 MFMAs per product; operands read from HBM:
        # per Linear its G tensor once per 128-feature k-block of A and its A tensor once (bytes from the layer widths)
        f_wg = 2 * (566_400 + 100_480)
        roof = {"bound": "hbm", "kernel": "k_wgrad_h (one grouped launch: dW = G^T A of all 162 Linears + the time-table one-hot GEMM)",
                "avg_launch_ms": wg, "share_of_step": wg / (dt / steps * 1e3),
                "algorithmic_tflops": f_wg * B / (wg * 1e-3) / 1e12,
                "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "phase_ms": {"forward": fwd, "activation_backward": bwd, "column_sums": cs, "weight_gradients": wg, "reduce_and_time_path": tail},
                "phase_note": "event marks on the caller's stream; from 32 768 rows on the column sums and the time-path backward run on the side "
                              "stream beside the weight gradients, so `column_sums` is only the max|G| reduce and `reduce_and_time_path` only "
                              "the closing reduce (DESIGN.md 5)"}
        # algorithmic bytes of the launch: every saved activation (A operand: x, h1, h2 per block + Linear inputs) and every
        # gradient tensor (G operand) read once, float32
        blocks = [(128, 128, 0)] * 2 + [(64, 64, 0)] * 2 + [(32, 32, 0)] * 2 + [(16, 16, 0)] * 2 + [(8, 8, 0)] * 4 + \
                 [(8, 8, 8)] * 3 + [(16, 16, 16)] * 3 + [(32, 32, 32)] * 3 + [(64, 64, 64)] * 3 + [(128, 128, 128)] * 3
        # From 1 024 row tiles on, the weight gradients of the last six residual blocks (3 x 128-wide, 3 x 64-wide up blocks) run as two
        # early launches on a side stream beside the activation-gradient chain (dsg_train_step); the timed launch is the rest
        early = 6 if (B + 31) // 32 >= 1024 else 0
        if early:
            blocks = blocks[:-early]
            roof["kernel"] = ("k_wgrad_h, the step's tail: dW = G^T A of every Linear except the last six residual blocks' (those run on a "
                              "side stream beside the activation-gradient chain).  `avg_launch_ms` = the phase between the end of the chain and "
                              "the closing reduce on the caller's stream (the blocks' units, then the plain Linears'); the side stream runs the "
                              "time-table units, the column sums and the time-path backward beside it (DESIGN.md 5)")
            roof["algorithmic_tflops"] = None
        per_row = 0
        for n, i0, i1 in blocks:
            per_row += 4 * ((i0 + i1) + 2 * n + 3 * n + 80)        # A: x, h1, h2, cond ; G: dh1, dh2, dout
        lins = [(80, 128), (128, 64), (64, 32), (32, 16), (16, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 80)]
        per_row += sum(4 * (k + n) for k, n in lins)
        roof["algorithmic_bytes_per_launch"] = per_row * B
        roof["achieved"] = per_row * B / (wg * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / PEAK_HBM_GBS
        # counter traffic, imported like the sampling leg's (labelled; separate --pmc passes of tools/pmc_train.sh at 32 768 rows)
        ttp = os.path.join(ROOT, "profiles", "traffic_train.json")
        if os.path.exists(ttp) and B == 32768:
            tt = json.load(open(ttp))
            kw = tt["kernels"].get("k_wgrad_h", {})
            roof["traffic"] = kw.get("bytes_per_launch")
            roof["traffic_per_step_all_k_wgrad_h_launches"] = kw.get("bytes_per_step")
            roof["step_traffic_bytes_listed_kernels"] = tt.get("bytes_per_step_listed_kernels")
            roof["traffic_source"] = "imported, not measured in this run: profiles/traffic_train.json <- " + tt.get("summary", "") + \
                                     " (mean k_wgrad_h launch, five per step; 2 x FETCH_SIZE + WRITE_SIZE, fabric side)"
    return {"roofline": roof, "samples_per_s": sps, "ms_per_step": dt / steps * 1e3, "host_ms_per_step": t_host / steps * 1e3,
            "graph": ("one captured HIP graph per step (train.StepGraph: draws + forward + backward + Adam + zero_grad + re-pack)" if sg is not None
                      else (graph_note or "eager launches")),
            "batch_per_gpu": B, "global_batch": world * B,
            "steps": steps, "T": 20, "final_loss": float(loss.detach()), "draws": "device Philox per rank (dsg_train_step_seeded, seed 1000 + rank)", "achieved_tflops": sps / world * f_train / 1e12,
            "frac_f32_mfma": sps / world * f_train / 1e12 / PEAK_F32_TFLOPS, "grad_bucket_bytes": int(ddpm.grad_bucket.numel()) * 4,
            "collective": ("none (1 GPU)" if world == 1 else
                           "one all_reduce(AVG) per step over the flat bucket (RCCL)" if dist.get_backend() == "nccl" else
                           f"one all_reduce(SUM) + divide per step over the flat bucket ({dist.get_backend()}: one-GPU plumbing test)"),
            "ranks_seen": ev["ranks_seen"], "per_rank_samples_per_s": [B * r for r in ev["per_rank_steps_per_s"]],
            "bucket_checksum_equal": bool(eq), "bucket_checksum": csum}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536, help="rows per GPU")
    ap.add_argument("--omega", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["split_f16", "f32"], default="split_f16")
    ap.add_argument("--train-batch", type=int, default=32768,
                    help="training rows per GPU: BASELINE config 4 is a global batch of 262144 over 8 GPUs")
    ap.add_argument("--train-steps", type=int, default=30)
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-train-graph", action="store_true", help="time the eager training loop instead of the captured step graph (one GPU)")
    ap.add_argument("--no-f32-exact", action="store_true", help="skip the exact-float32 re-run of the same K steps")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE config-2 sub-record (MSR-3c, 8192 rows, T = 1000)")
    ap.add_argument("--warm-seconds", type=float, default=0.3, help="untimed clock warm-up before the timed steps")
    ap.add_argument("--repeats", type=int, default=7, help="consecutive timed K-step sample() calls; `value` is the median (each is exactly K steps)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)                      # never returns; nothing has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    ndev = torch.cuda.device_count()
    if local >= ndev and not os.environ.get("DSG_BENCH_BACKEND"):
        raise SystemExit(f"LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    local_dev = local % ndev                 # (only DSG_BENCH_BACKEND=gloo lets several ranks share a GPU: plumbing smoke test on one GPU)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("DSG_BENCH_BACKEND", "nccl")     # "nccl" = RCCL over xGMI; "gloo" only for the one-GPU smoke test
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    K, W, B = a.steps, max(a.warmup, 1), a.batch
    ddpm_k = build_model(dev, K)
    ddpm_w = build_model(dev, W)
    ddpm_w.model = ddpm_k.model          # one denoiser (one native handle, one captured graph) for both schedules
    ddpm_k.model.set_precision(a.precision)
    g = torch.Generator().manual_seed(rank)
    cond = torch.rand(B, 80, generator=g).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ddpm_w.sample(cond, a.omega, seed=1)  # W untimed warm-up steps (also builds tables, captures the step graph)
    # a box that was idle ramps its clocks over some tens of milliseconds: keep the GPU busy (untimed, same W-step call) for
    # ~0.3 s before the timed region so that the K timed steps are measured at the sustained clock
    torch.cuda.synchronize()
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < a.warm_seconds:
        ddpm_w.sample(cond, a.omega, seed=1)
        torch.cuda.synchronize()
    # the K-step schedule's coefficient table (4 floats per step, a function of the registered schedule buffers only) is built
    # on an object's first call and cached on it; the timed object is fresh only because the warm-up uses its own W-step schedule
    ddpm_k._coef_table()
    box = box_calibrate()                 # fixed MFMA / copy probes on this box, right before the timed region (also keeps the clock up)
    from diffsg_amd import parallel as par
    # R consecutive timed calls of exactly K steps each, every one between barrier + synchronize; `value` is the MEDIAN call (the
    # first call of a process and a box's clock excursions fall out of it), min / max ride along in `repeats`
    evs = []
    for _ in range(max(a.repeats, 1)):
        barrier()
        t0 = time.perf_counter()
        y0 = ddpm_k.sample(cond, a.omega, seed=2)   # exactly K timed steps
        barrier()
        evs.append(par.run_evidence(dev, time.perf_counter() - t0, K))    # max over ranks of this call
    evs_sorted = sorted(evs, key=lambda e: e["max_seconds"])
    ev = evs_sorted[len(evs_sorted) // 2]
    dt = ev["max_seconds"]
    rep_ms = [e["max_seconds"] / K * 1e3 for e in evs]
    assert torch.isfinite(y0).all()
    box_after = box_calibrate()

    # the same K steps on the exact-float32 kernels (v_mfma_f32_32x32x2_f32), so that a strict-fp32 figure always sits beside the
    # split-f16 one.  Untimed warm-up first: the precision switch drops the captured step graphs.
    f32_exact = None
    if a.precision == "split_f16" and not a.no_f32_exact:
        ddpm_k.model.set_precision("f32")
        ddpm_w.sample(cond, a.omega, seed=1)
        barrier()
        t1 = time.perf_counter()
        y1 = ddpm_k.sample(cond, a.omega, seed=2)
        barrier()
        ev1 = par.run_evidence(dev, time.perf_counter() - t1, K)
        ddpm_k.model.set_precision("split_f16")
        ddpm_w.sample(cond, a.omega, seed=1)          # back on the split path (re-captures its graphs) before anything else is timed
        torch.cuda.synchronize()
        f32_exact = {"value": world * K / ev1["max_seconds"], "unit": "steps/s", "ms_per_step": ev1["max_seconds"] / K * 1e3,
                     "dtype": "f32 (v_mfma_f32_32x32x2_f32, every GEMM exact float32)",
                     "max_rel_diff_vs_split": float((y1 - y0).abs().max() / y1.abs().max())}

    train = None
    if not a.no_train:
        train = train_leg(dev, dist, rank, world, a.train_batch, a.train_steps, barrier, use_graph=not a.no_train_graph)

    if rank == 0:
        # roofline of the dominant kernel: eager re-run of the same K steps with HIP events around every launch
        torch.cuda.synchronize()
        te = time.perf_counter()
        ddpm_k.sample(cond, a.omega, seed=2, profile=True)
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - te) / K * 1e3
        prof = ddpm_k.op_profile()
        # operators that ran inside the previous operator's launch (fused narrow run, block + Linear pairs) read only the
        # empty event pair (~5 us; the smallest real launch is ~14 us): fold their algorithmic work into the launch that did it
        merged = []
        for r in prof:
            if merged and r[4] > 0 and r[3] < 8e-3 * r[4]:
                m = merged[-1]
                merged[-1] = (m[0] + "+" + r[0].split(".")[-2] + "." + r[0].split(".")[-1] if "." in r[0] else m[0] + "+" + r[0],
                              m[1] + r[1], m[2] + r[2], m[3], m[4])
            else:
                merged.append(r)
        prof = merged
        # dominant kernel = the operator shape with the largest algorithmic FLOP count (the proj_dim-wide up blocks)
        pure = [r for r in prof if "+" not in r[0]]
        name, fl, by, ms, calls = max(pure, key=lambda r: (r[1], r[3]))
        dom = [r for r in pure if r[1] == fl and r[0].split(".")[0] == name.split(".")[0]]
        ms_sum, n_calls = sum(r[3] for r in dom), sum(r[4] for r in dom)
        avg_ms = ms_sum / n_calls
        ach = fl * B / (avg_ms * 1e-3) / 1e12            # algorithmic (float32-equivalent) TFLOP/s
        split = a.precision == "split_f16"
        mfma_x = 3.0 if split else 1.0                  # executed MFMA FLOP per algorithmic FLOP
        unit_peak = PEAK_F16_TFLOPS if split else PEAK_F32_TFLOPS
        step_ms = dt / K * 1e3
        traffic = valu_per_mfma = tsrc = step_bytes = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic, valu_per_mfma = tj.get("hbm_bytes_per_launch"), tj.get("valu_per_mfma")
            step_bytes = (tj.get("step_traffic") or {}).get("bytes_per_step")
            tsrc = "imported, not measured in this run: profiles/traffic.json <- " + tj.get("summary", "profiles/") + " (kernel " + tj.get("kernel", "?") + ")"
        large = 2 * ((B + 31) // 32) > 512
        panel = 2 * ((B + 31) // 32) >= 768
        kname = (("k_panel128_h<linear-shortcut, half panels> (persistent, two 4-wave workgroups per CU, 16 KiB weight panels through LDS, operand preparation "
                  "interleaved with the MFMA stream)" if panel else
                  "k_wide128_h<linear-shortcut> (4 tiles per workgroup, weight planes through an LDS ring)") if large
                 else "k_resblock_c<128,linear-shortcut>") if split else "k_resblock<128,linear-shortcut>"
        out = {
            "metric": "ddpm_reverse_sample_steps_per_sec_msr80c", "value": world * K / dt, "unit": "steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": step_ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32(2xf16-split MFMA, f32 accumulate)" if a.precision == "split_f16" else "f32", "data": "synthetic",
            "config": {"workload": f"MSR-80c CFG reverse sampling, batch {B} x D=C=80 per GPU, UNet1D(proj 128, dims "
                                   f"(64,32,16,8), n_blocks 2), omega={a.omega:g}, device Philox noise; rows sharded, no collective",
                       "batch_per_gpu": B, "solution_dim": 80, "parallelism": f"rows x{world}"},
            "row_steps_per_s": world * K * B / dt,
            # `value` = K / the median of R consecutive timed K-step calls; the spread of the calls and this box's rates on four fixed
            # probes ride along
            "repeats": {"n": len(rep_ms), "ms_per_step_min": min(rep_ms), "ms_per_step_median": step_ms, "ms_per_step_max": max(rep_ms),
                        "ms_per_step_all": [round(v, 4) for v in rep_ms]},
            "box": {"before": box, "after": box_after,
                    "probe": "dsg_box_calibrate, median of five 3-10 ms launches each, run right before and right after the timed calls: mfma_tflops = "
                             "dependent v_mfma_f32_32x32x16_f16 on every SIMD; mix_gslots = the same with 6 vector instructions behind every MFMA "
                             "(1e9 slots/s); copy_gbs = 256 MiB float4 device copy, read + written bytes; panel_gslots = a frozen miniature of the "
                             "128-wide panel kernels' load profile (LDS-DMA weight stream + activation stream + LDS reads + MFMA + vector "
                             "instructions), 1e9 MFMA slots/s",
                    "note": "round 5, five boxes: the four probes agree within 2 % (panel_gslots 5 %) between boxes whose `value` differs by 8 % "
                            "-- the spread of the pool is NOT the matrix-pipe clock, the vector rate or the copy bandwidth, and no probe normalises "
                            "it; compare lines through `repeats` (within-box spread ~1 %) and `roofline.avg_launch_ms`, and trees through same-box "
                            "A-B runs (tools/tree_ab.sh; DESIGN.md 8)"},
            "roofline": {"bound": "mfma", "kernel": f"{kname} ({len(dom)} launches/step: {', '.join(r[0] for r in dom)})",
                         "achieved": mfma_x * ach, "peak": unit_peak, "unit": "TFLOP/s", "frac": mfma_x * ach / unit_peak,
                         "mfma_unit": "v_mfma_f32_32x32x16_f16, 3 per float32 product (hi*hi + hi*lo + lo*hi)" if split else "v_mfma_f32_32x32x2_f32",
                         "algorithmic_tflops": ach, "frac_f32_equiv": ach / PEAK_F32_TFLOPS,
                         "traffic": traffic, "valu_per_mfma": valu_per_mfma, "traffic_source": tsrc,
                         "traffic_note": (tj.get("note") if os.path.exists(tpath) else None),
                         "algorithmic_bytes_per_launch": by * B, "avg_launch_ms": avg_ms, "flop_per_launch": fl * B,
                         "executed_mfma_flop_per_launch": mfma_x * fl * B, "share_of_step": ms_sum / sum(r[3] for r in prof)},
            "step_roofline": {"f_alg_per_row": F_ALG, "achieved_tflops": F_ALG * B / (step_ms * 1e-3) / 1e12,
                              "frac_mfma_unit": mfma_x * F_ALG * B / (step_ms * 1e-3) / 1e12 / unit_peak,
                              "frac_f32_equiv": F_ALG * B / (step_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                              "achieved_gbs_alg": BYT_ALG * B / (step_ms * 1e-3) / 1e9,
                              "frac_hbm": BYT_ALG * B / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                              # the binding memory-side quantity: fabric traffic of the step's large kernels from the committed PMC summary
                              # (imported like roofline.traffic; Infinity-Cache hits included), against the timed step of THIS run
                              "traffic_bytes": step_bytes, "traffic_source": tsrc,
                              "traffic_gbs": (step_bytes / (step_ms * 1e-3) / 1e9) if step_bytes else None,
                              "frac_hbm_traffic": (step_bytes / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if step_bytes else None},
            # per-operator HIP-event times of a separate EAGER run of the same K steps: an event pair adds a few microseconds to every
            # launch and the launches do not overlap their tails as they do inside the replayed graph, so these sum ABOVE ms_per_step;
            # `eager_profile_ms_per_step` is that run's own wall time per step (upper bounds, for ratios between operators)
            "op_ms_per_step": {r[0]: r[3] / K for r in prof}, "eager_profile_ms_per_step": eager_ms,
            "ranks_seen": ev["ranks_seen"], "per_rank_steps_per_s": ev["per_rank_steps_per_s"],
        }
        # The driver's parser keeps `roofline` and `cpu_baseline` whole and only the NAMES of other top-level keys: what a
        # strict reader needs sits INSIDE `roofline` -- the exact-float32 rate against SURVEY 8(d)'s 973-steps/s fp32 roof, the whole step
        # against the unit it runs on, its traffic, the training step, and every other BASELINE config under `roofline.configs`.
        rl = out["roofline"]
        rl["step_ms"] = step_ms
        rl["step_frac_mfma_unit"] = out["step_roofline"]["frac_mfma_unit"]
        rl["step_frac_fp32_roof"] = out["step_roofline"]["frac_f32_equiv"]
        rl["step_traffic_bytes"] = step_bytes
        rl["step_frac_hbm_traffic"] = out["step_roofline"]["frac_hbm_traffic"]
        # the same traffic against what THIS box moves in a plain device copy (dsg_box_calibrate, read + written bytes per second): the step's large
        # kernels stream 33 x the algorithmic bytes (every block's activations round-trip memory) and sit at ~0.75 of that rate
        copy_gbs = (box.get("copy_gbs") if isinstance(box, dict) else None)
        rl["box_copy_gbs"] = copy_gbs
        rl["step_traffic_gbs"] = out["step_roofline"]["traffic_gbs"]
        rl["step_frac_box_copy_bw"] = (out["step_roofline"]["traffic_gbs"] / copy_gbs) if (copy_gbs and out["step_roofline"]["traffic_gbs"]) else None
        fp32_roof_steps = PEAK_F32_TFLOPS * 1e12 / (F_ALG_SURVEY * B)        # 973 steps/s at 65 536 rows (cond GEMMs counted)
        rl["fp32_roof_steps_per_s"] = fp32_roof_steps
        rl["split_frac_fp32_roof"] = (K / dt) / fp32_roof_steps
        if f32_exact is not None:
            out["f32_exact"] = f32_exact
            rl["f32_exact_steps_per_s"] = f32_exact["value"] / world
            rl["f32_exact_frac_fp32_roof"] = f32_exact["value"] / world / fp32_roof_steps
            rl["f32_exact_max_rel_diff_vs_split"] = f32_exact["max_rel_diff_vs_split"]
        rl["configs"] = {}
        if train is not None:
            out["train"] = train
            tr_roof = train.get("roofline") or {}
            rl["configs"]["msr80_train"] = {
                "workload": f"MSR-80c training (BASELINE config 5 shape), {train['batch_per_gpu']} rows per GPU x {world} GPU(s), T=20, Adam + re-pack"
                            + ("" if world == 1 else ", one all-reduce of the flat gradient bucket per step"),
                "samples_per_s": train["samples_per_s"], "ms_per_step": train["ms_per_step"], "algorithmic_tflops_per_gpu": train["achieved_tflops"],
                "frac_fp32_roof": train["frac_f32_mfma"], "phase_ms": tr_roof.get("phase_ms"), "wgrad_tail_ms": tr_roof.get("avg_launch_ms"),
                "host_ms_per_step": train.get("host_ms_per_step"), "graph": train.get("graph"),
                "bucket_checksum_equal": train["bucket_checksum_equal"], "ranks_seen": train["ranks_seen"]}
        if world == 1 and not a.no_other_configs:
            out["config2"] = config2_leg(dev)
            c2 = out["config2"]
            f2 = 2 * 2 * 546_688
            rl["configs"]["msr3_config2"] = {"workload": c2["workload"], "steps_per_s": c2["value"], "ms_per_step": c2["ms_per_step"],
                                             "row_steps_per_s": c2["row_steps_per_s"], "f_alg_per_row_step": f2,
                                             "frac_fp32_roof": f2 * c2["row_steps_per_s"] / 1e12 / PEAK_F32_TFLOPS, "finite": c2["finite"]}
            rl["configs"].update(configs_leg(dev, split))
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
