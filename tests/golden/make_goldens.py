#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference itself.

Runs ONLY in the build container (needs /root/reference); the outputs are
committed.  Nothing here is imported by tests at run time except weights.py.

    python tests/golden/make_goldens.py            # all groups
    python tests/golden/make_goldens.py G2 G4      # selected groups

Groups follow SURVEY.md section 8(c): G1 schedule, G2 denoiser forward (+ per
module taps), G3 DDPM.forward loss/grads, G4 DDPM.sample trajectories (NU
checkpoint + synthetic), G5 decoders/evaluators, G6 loaders on CSV slices,
G7 state-dict layout + EMA, G8 the MSR label generator, G9 the CO self-check harness,
G10 DDPM.sample at T = 1000 (BASELINE config 2's schedule length), G11 the CO label generator.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
os.chdir(os.path.join(REF, "ddpm_opt"))

from ddpm_opt.diffusion import generate_cosine_schedule, init_weights  # noqa: E402
from ddpm_opt.UNetCF import UNet1D, ResidualBlock  # noqa: E402
import ddpm_opt.classifier_free_MSR as RMSR  # noqa: E402
import ddpm_opt.classifier_free_CO as RCO  # noqa: E402
import ddpm_opt.classifier_free_NU as RNU  # noqa: E402
from weights import CONFIGS, synth_weights  # noqa: E402

torch.set_num_threads(8)


def ref_unet(cfg):
    n = len(cfg["dims"])
    return UNet1D(input_dim=cfg["input_dim"], proj_dim=cfg["proj_dim"], cond_dim=cfg["cond_dim"],
                  dims=cfg["dims"], is_attn=(False,) * n, middle_attn=False, n_blocks=cfg["n_blocks"])


def load_synth(model, seed, flavour):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth_weights(shapes, seed, flavour)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return shapes


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


# ---------------------------------------------------------------- G1
def g1():
    out = {}
    for T in (4, 20, 400, 500, 1000):
        betas = generate_cosine_schedule(T)
        out[f"T{T}_betas_f64"] = betas
        alphas = 1.0 - betas
        m = RMSR.DDPM(T, torch.nn.Identity(), 3, 10.0, alphas, "cpu", (1, 3))
        for k, v in m.state_dict().items():
            if not k.startswith("ema."):
                out[f"T{T}_{k}"] = v.numpy()
    save("g1_schedule.npz", **out)


# ---------------------------------------------------------------- G2
def module_taps(model):
    """Forward hooks on every direct child of down/up + middle + time_emb + feature_proj."""
    taps = {}

    def hook(name):
        def f(_m, _inp, out):
            taps[name] = out.detach().clone().numpy()
        return f
    hs = [model.time_emb.register_forward_hook(hook("time_emb")),
          model.feature_proj.register_forward_hook(hook("feature_proj")),
          model.middle.register_forward_hook(hook("middle"))]
    for i, m in enumerate(model.down):
        hs.append(m.register_forward_hook(hook(f"down.{i}")))
    for i, m in enumerate(model.up):
        hs.append(m.register_forward_hook(hook(f"up.{i}")))
    return taps, hs


def g2():
    B = 48
    for name, cfg in CONFIGS.items():
        for flavour, seed in (("trained", 11), ("init", 12)):
            if flavour == "init" and name not in ("msr80", "tiny"):
                continue
            model = ref_unet(cfg)
            load_synth(model, seed, flavour)
            rs = np.random.RandomState(100 + seed)
            D, C = cfg["input_dim"], cfg["cond_dim"]
            x = torch.from_numpy(rs.standard_normal((B, D)).astype(np.float32))
            cond = torch.from_numpy(rs.uniform(0, 1, (B, C)).astype(np.float32))
            out = dict(x=x.numpy(), cond=cond.numpy())
            want_taps = name in ("msr80", "nu3", "tiny") and flavour == "trained"
            # (a) training-like: per-row t = ts/T, random mask;  (b)/(c) sampling-like: uniform t, mask 0 / 1
            ts = torch.from_numpy(rs.randint(0, 20, (1, B)).astype(np.int64))
            mask = torch.from_numpy((rs.uniform(0, 1, (B, 1)) < 0.7).astype(np.float32))
            with torch.no_grad():
                if want_taps:
                    taps, hs = module_taps(model)
                out["a_ts"] = ts.numpy(); out["a_T"] = np.int64(20); out["a_mask"] = mask.numpy()
                out["a_eps"] = model(x, ts / 20, cond, mask).numpy()
                if want_taps:
                    for k, v in taps.items():
                        out[f"a_tap.{k}"] = v
                    for h in hs:
                        h.remove()
                t7 = torch.full((1, B), 7, dtype=torch.int64) / 20
                out["b_step"] = np.int64(7)
                out["b_eps"] = model(x, t7, cond, torch.zeros(B, 1)).numpy()
                out["c_eps"] = model(x, t7, cond, torch.ones(B, 1)).numpy()
            save(f"g2_unet_{name}_{flavour}.npz", **out)


# ---------------------------------------------------------------- G3
def g3():
    T = 20
    for name, B in (("tiny", 48), ("nu3", 48), ("msr80", 40)):
        cfg = CONFIGS[name]
        model = ref_unet(cfg)
        alphas = 1.0 - generate_cosine_schedule(T)
        D, C = cfg["input_dim"], cfg["cond_dim"]
        ddpm = RMSR.DDPM(T, model, D, 10.0, alphas, torch.device("cpu"), (1, D), None, 0.1, 0.9999, 10, 5, False)
        load_synth(ddpm.model, 21, "trained")
        rs = np.random.RandomState(300)
        y = torch.from_numpy(rs.uniform(0, 1, (B, D)).astype(np.float32))
        cond = torch.from_numpy(rs.uniform(0, 1, (B, C)).astype(np.float32))
        seed = 4321
        torch.manual_seed(seed)
        import random
        random.seed(0)  # keeps the debug print (MSR.py:110) quiet: first random() = 0.844
        loss = ddpm(y, cond)
        loss.backward()
        # the three draws, replayed in the reference's order (MSR.py:101,102,107)
        torch.manual_seed(seed)
        ts = torch.randint(low=0, high=T, size=(1, B))
        noise = torch.randn_like(y)
        mask = torch.bernoulli(torch.fill(torch.zeros(B), 1 - 0.1))[:, None]
        out = dict(y=y.numpy(), cond=cond.numpy(), ts=ts.numpy(), noise=noise.numpy(), mask=mask.numpy(),
                   loss=loss.detach().numpy(), T=np.int64(T))
        for k, p_ in ddpm.model.named_parameters():
            g = p_.grad.detach().numpy()
            if name == "tiny":
                out["grad." + k] = g
            else:
                out["gradnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                out["gradhead." + k] = g.reshape(-1)[:16].copy()
        save(f"g3_loss_{name}.npz", **out)


# ---------------------------------------------------------------- G4
def replay_sample_noise(seed, B, D, T):
    torch.manual_seed(seed)
    y_T = torch.randn(B, 1, D).squeeze()
    z = {}
    for i in range(T - 1, -1, -1):
        if i > 1:
            z[i] = torch.randn(B, 1, D).squeeze()
    return y_T, z


def g4():
    # (1) the shipped NU checkpoint on the first 512 test rows
    T = 20
    X_train, Y_train, X_test, Y_test, R_test, ccfg = RNU.nu_data_load("../datasets/3u_18mW_10000samples.csv", 400, 400)
    K, P_sum = ccfg["K"], ccfg["P_sum"]
    model = ref_unet(CONFIGS["nu3"])
    alphas = 1.0 - generate_cosine_schedule(T)
    ddpm = RNU.DDPM(T, model, K, P_sum, alphas, torch.device("cpu"), (1, 2 + K), ccfg, 0.1, 0.9999, 10, 5, False)
    sd = torch.load("../ckpts/ddpm_nu_3u.pt", map_location="cpu")
    ddpm.load_state_dict(sd)
    B = 512
    cond = torch.tensor(X_test[:B], dtype=torch.float32)
    out = dict(cond=cond.numpy(), y_test=np.asarray(Y_test[:B], dtype=np.float32), T=np.int64(T),
               P_sum=np.float64(P_sum))
    for k, v in sd.items():
        if k.startswith("model."):
            out["w." + k[len("model."):]] = v.numpy()
    seed = 1234
    y_T, z = replay_sample_noise(seed, B, 2 + K, T)
    out["y_T"] = y_T.numpy()
    out["z"] = np.stack([z[i].numpy() for i in range(T - 1, 1, -1)])  # order: i = T-1 .. 2
    with torch.no_grad():
        for omega in (0.0, 1.0, 500.0):
            torch.manual_seed(seed)
            ddpm.record_denoise_path = False
            y0 = ddpm.sample(cond, omega)
            out[f"om{omega:g}_y0"] = y0.numpy()
        # per-step eps for omega=1: eps_i_record is kept raw by the record branch (MSR.py:153-154); the NU class's
        # own record branch references undefined globals (NU.py:176), so the byte-identical MSR class records it.
        rec = RMSR.DDPM(T, ref_unet(CONFIGS["nu3"]), K, P_sum, alphas, torch.device("cpu"), (1, 2 + K), ccfg)
        rec.load_state_dict(sd)
        torch.manual_seed(seed)
        rec.record_denoise_path = True
        rec.sample(cond, 1.0)
        out["om1_eps_steps"] = rec.eps_i_record.reshape(B, T, 2 + K).transpose(1, 0, 2).astype(np.float32)
        # known answer: less ratio at omega=500 on these rows (0.91359)
        y0 = torch.from_numpy(out["om500_y0"])
        Xs = cond.clone()
        for i in range(K):
            Xs[:, 2 * i] *= 400
            Xs[:, 2 * i + 1] *= 400
        Yd = RNU.custom_decoder(y0, 400, 400, P_sum)
        Yt = torch.tensor(Y_test[:B], dtype=torch.float32)
        Yt[:, 0] *= 400; Yt[:, 1] *= 400; Yt[:, 2:] *= P_sum
        pr, tr = RNU.rate_calc(Yd, Xs), RNU.rate_calc(Yt, Xs)
        out["om500_pred_rate"] = pr.numpy(); out["om500_true_rate"] = tr.numpy()
        out["om500_less_ratio"] = np.float64(torch.sum(pr) / torch.sum(tr))
        print("NU less ratio (omega=500, 512 rows):", out["om500_less_ratio"])
        # fp64 evaluation of the same trajectory -> error budget
        d64 = RNU.DDPM(T, ref_unet(CONFIGS["nu3"]), K, P_sum, alphas, torch.device("cpu"), (1, 2 + K), ccfg)
        d64.load_state_dict(sd)
        d64 = d64.double()
        for omega in (0.0, 1.0, 500.0):
            y = y_T.double()
            c64 = cond.double()
            for i in range(T - 1, -1, -1):
                t = (torch.full((1, B), i) / T).double()
                e0 = d64.model(y, t, c64, torch.zeros(B, 1, dtype=torch.float64))
                e1 = d64.model(y, t, c64, torch.ones(B, 1, dtype=torch.float64))
                nz = z[i].double() if i > 1 else 0
                e = (1 + omega) * e1 - omega * e0
                y = (y - d64.betas[i] / d64.sqrt_one_minus_alphas_cumprod[i] * e) * d64.reciprocal_sqrt_alphas[i] \
                    + (1.0 - d64.alphas_cumprod[i - 1 if i - 1 >= 0 else 0]) / (1.0 - d64.alphas_cumprod[i]) * nz
                if i > T - 5:
                    y = (y - torch.mean(y)) / torch.sqrt(torch.var(y))
            out[f"om{omega:g}_y0_f64"] = y.numpy()
    save("g4_sample_nu_ckpt.npz", **out)

    # (2) synthetic weights, several configs, short T (T=8 also exercises i>T-5 and the i<=1 no-noise rule)
    for name, B, T in (("tiny", 40, 8), ("msr80", 40, 6), ("msr3", 33, 6), ("co3", 40, 6), ("tiny", 5, 3)):
        cfg = CONFIGS[name]
        D, C = cfg["input_dim"], cfg["cond_dim"]
        alphas = 1.0 - generate_cosine_schedule(T)
        ddpm = RMSR.DDPM(T, ref_unet(cfg), D, 10.0, alphas, torch.device("cpu"), (1, D), None)
        load_synth(ddpm.model, 31, "trained")
        rs = np.random.RandomState(400)
        cond = torch.from_numpy(rs.uniform(0, 1, (B, C)).astype(np.float32))
        seed = 77
        y_T, z = replay_sample_noise(seed, B, D, T)
        out = dict(cond=cond.numpy(), y_T=y_T.numpy(), T=np.int64(T),
                   z=(np.stack([z[i].numpy() for i in range(T - 1, 1, -1)]) if T > 2 else np.zeros((0, B, D), np.float32)))
        with torch.no_grad():
            for omega in (0.0, 1.0, 3.0):
                torch.manual_seed(seed)
                out[f"om{omega:g}_y0"] = ddpm.sample(cond, omega).numpy()
        if name == "tiny" and T == 8:
            torch.manual_seed(seed)
            ddpm.record_denoise_path = True
            with torch.no_grad():
                y0 = ddpm.sample(cond, 1.0)
            save("g4_record_tiny_T8.npz", y_i_record=ddpm.y_i_record.astype(np.float32), eps_i_record=ddpm.eps_i_record.astype(np.float32),
                 y0=y0.numpy())
            ddpm.record_denoise_path = False
        save(f"g4_sample_{name}_T{T}.npz", **out)


# ---------------------------------------------------------------- G5
def g5():
    rs = np.random.RandomState(500)
    out = {}
    y = torch.from_numpy(rs.standard_normal((64, 3)).astype(np.float32) * 3)
    g = torch.from_numpy(rs.uniform(0.5, 2.5, (64, 3)).astype(np.float32))
    out["msr_y"] = y.numpy(); out["msr_gain"] = g.numpy()
    dec = RMSR.custom_decoder(y)
    out["msr_dec"] = dec.numpy()
    out["msr_rate"] = torch.sum(torch.log2(1.0 + 10.0 * dec * g), dim=1).numpy()
    y = torch.from_numpy(rs.standard_normal((64, 3)).astype(np.float32) * 8)
    y[5] = -11.0
    X = torch.from_numpy(rs.uniform(0.05, 4.0, (64, 9)).astype(np.float32))
    out["co_y"] = y.numpy(); out["co_X"] = X.numpy()
    dec = RCO.customized_real_decoder(y)
    out["co_dec"] = dec.numpy()
    out["co_cost"] = RCO.cost_calc(X, dec).numpy()
    y = torch.from_numpy(rs.standard_normal((64, 5)).astype(np.float32) * 2)
    X = torch.from_numpy(rs.uniform(0, 400, (64, 6)).astype(np.float32))
    out["nu_y"] = y.numpy(); out["nu_X"] = X.numpy()
    dec = RNU.custom_decoder(y, 400, 400, 18.0)
    out["nu_dec"] = dec.numpy()
    out["nu_rate"] = RNU.rate_calc(dec, X).numpy()
    save("g5_decoders.npz", **out)


# ---------------------------------------------------------------- G6
def g6():
    """Loader outputs on 200-row slices of the shipped CSVs (the slices are committed as data fixtures)."""
    import pandas as pd
    ddir = os.path.join(HERE, "data")
    os.makedirs(ddir, exist_ok=True)
    out = {}
    n = 200
    for src, dst in (("3c_10w_10000samples.csv", f"3c_10w_{n}samples.csv"),
                     ("3u_18mW_10000samples.csv", f"3u_18mW_{n}samples.csv"),
                     ("3nodes_2000samples_ood.csv", f"3nodes_{n}samples_ood.csv")):
        df = pd.read_csv(os.path.join(REF, "datasets", src), header=None)
        df.iloc[:n].to_csv(os.path.join(ddir, dst), header=False, index=False, float_format="%.17g")
    Xtr, Ytr, Xte, Yte, cfg = RMSR.msr_data_load(os.path.join(ddir, f"3c_10w_{n}samples.csv"))
    out.update(msr_Xtr=Xtr, msr_Ytr=Ytr, msr_Xte=Xte, msr_Yte=Yte,
               msr_cfg=np.array([cfg["M"], cfg["W"], cfg["scaler_min"], cfg["scaler_max"]], dtype=np.float64))
    Xtr, Ytr, Xte, Yte, Rte, cfg = RNU.nu_data_load(os.path.join(ddir, f"3u_18mW_{n}samples.csv"), 400, 400)
    out.update(nu_Xtr=Xtr, nu_Ytr=Ytr, nu_Xte=Xte, nu_Yte=Yte, nu_Rte=Rte,
               nu_cfg=np.array([cfg["K"], cfg["P_sum"]], dtype=np.float64))
    Xtr, Ytr, Xte, Yte, cfg = RCO.co_data_load(os.path.join(ddir, f"3nodes_{n}samples_ood.csv"))
    out.update(co_Xtr=Xtr, co_Ytr=Ytr, co_Xte=Xte, co_Yte=Yte,
               co_cfg=np.array([cfg["scaler_min"], cfg["scaler_max"]], dtype=np.float64))
    # full-file facts recorded as numbers only
    Xtr, Ytr, Xte, Yte, cfg = RMSR.msr_data_load("../datasets/3c_10w_10000samples.csv")
    out["msr_full_shapes"] = np.array([*Xtr.shape, *Ytr.shape, *Xte.shape, *Yte.shape])
    out["msr_full_cfg"] = np.array([cfg["M"], cfg["W"], cfg["scaler_min"], cfg["scaler_max"]], dtype=np.float64)
    save("g6_loaders.npz", **out)


# ---------------------------------------------------------------- G7
def g7():
    layout = {}
    for name, cfg in CONFIGS.items():
        T = 20
        D = cfg["input_dim"]
        ddpm = RMSR.DDPM(T, ref_unet(cfg), D, 10.0, 1.0 - generate_cosine_schedule(T), torch.device("cpu"), (1, D), None)
        layout[name] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ddpm.state_dict().items()]
    with open(os.path.join(HERE, "g7_state_layout.json"), "w") as f:
        json.dump(layout, f)
    print("wrote g7_state_layout.json", {k: len(v) for k, v in layout.items()})
    # EMA: AveragedModel semantics on the tiny net, three updates
    from ddpm_opt.ema import ExponentialMovingAverage
    model = ref_unet(CONFIGS["tiny"])
    load_synth(model, 41, "trained")
    ema = ExponentialMovingAverage(model, 0.9)
    out = {}
    for step in range(3):
        with torch.no_grad():
            for p_ in model.parameters():
                p_.add_(0.01 * (step + 1))
        ema.update_parameters(model)
        out[f"step{step}_n"] = ema.n_averaged.numpy().copy()
        out[f"step{step}_feature_proj.weight"] = ema.module.feature_proj.weight.detach().numpy().copy()
        out[f"step{step}_norm.bias"] = ema.module.norm.bias.detach().numpy().copy()
    save("g7_ema.npz", **out)
    # seeded construction + init_weights: per-tensor float64 sums pin the construction order / RNG consumption
    out = {}
    for name in ("tiny", "nu3"):
        torch.manual_seed(5)
        m = ref_unet(CONFIGS[name])
        m.apply(init_weights)
        out[name + "_sums"] = np.array([float(v.double().sum()) for v in m.state_dict().values()])
        out[name + "_abs"] = np.array([float(v.double().abs().sum()) for v in m.state_dict().values()])
    save("g7_seeded_init.npz", **out)


def g8():
    """utils/dataset_generate.py:247-313: SUM_RATE_GEN (LRH gradient descent labels of the MSR problem), float64."""
    import contextlib, io
    from utils.dataset_generate import SUM_RATE_GEN, SUM_RATE_GRAD, alpha_calc
    out = {}
    for tag, n, M, W in (("m3", 40, 3, 10.0), ("m80", 48, 80, 20.0), ("m7", 33, 7, 5.0)):
        np.random.seed(1000 + M)
        with contextlib.redirect_stdout(io.StringIO()):
            gs, rates, schemes = SUM_RATE_GEN(sample_num=n, M=M, W=W)
        out[tag + "_gs"], out[tag + "_rates"], out[tag + "_schemes"] = gs, rates, schemes
        out[tag + "_W"] = np.array(W)
    # one alpha_calc step on its own (the piece with the sort)
    np.random.seed(7)
    gs = np.random.uniform(0.5, 2.5, size=(9, 12))
    sch = np.random.uniform(0.1, 1.0, size=(9, 12))
    grad = SUM_RATE_GRAD(gs, sch)
    out["step_gs"], out["step_schemes"], out["step_grad"], out["step_alpha"] = gs, sch, grad, alpha_calc(grad)
    save("g8_sum_rate_gen.npz", **out)


def g9():
    """classifier_free_CO.py:416-449 validation_data_gen (numpy draw order, split), and the decision-pattern accuracy rule
    of test_ddpm (:542-552) evaluated on fixed raw samples."""
    np.random.seed(321)
    Xtr, Ytr, Xte, Yte, cfg = RCO.validation_data_gen()
    out = dict(Xtr_head=Xtr[:16], Ytr_head=Ytr[:16], Xte_tail=Xte[-16:], Yte_tail=Yte[-16:],
               shapes=np.array([Xtr.shape, Ytr.shape, Xte.shape, Yte.shape]), sums=np.array([Xtr.sum(), Ytr.sum(), Xte.sum(), Yte.sum()]),
               sfn=np.array(cfg['sfn']), cfn=np.array(cfg['cfn']))
    g = torch.Generator().manual_seed(5)
    raw = torch.randn(200, 3, generator=g) * 3.0
    lab = torch.zeros(200, 3)
    lab[torch.arange(200), torch.randint(0, 3, (200,), generator=g)] = 1
    Y_pred = torch.softmax(raw, dim=1)
    pd_, td_ = torch.where(Y_pred > 0.1, 1, 0), torch.where(lab > 0.1, 1, 0)
    pc, tc = np.zeros(200, dtype=int), np.zeros(200, dtype=int)
    for i in range(3):
        pc = pc + (pd_[:, i] * (2 ** (3 - i - 1))).numpy()
        tc = tc + (td_[:, i] * (2 ** (3 - i - 1))).numpy()
    out.update(acc_raw=raw.numpy(), acc_lab=lab.numpy(), acc_hits=np.array(int(np.sum(np.where(tc == pc, 1, 0)))))
    save("g9_co_validation.npz", **out)


# ---------------------------------------------------------------- G10
def g10():
    """BASELINE config 2's schedule length: DDPM.sample at T = 1000 (MSR-3c, 16 rows), the reference's float32 result and a
    float64 evaluation of the same trajectory (the error budget of 2 000 chained float32 forwards)."""
    name, B, T, seed = "msr3", 16, 1000, 78
    cfg = CONFIGS[name]
    D, C = cfg["input_dim"], cfg["cond_dim"]
    alphas = 1.0 - generate_cosine_schedule(T)
    ddpm = RMSR.DDPM(T, ref_unet(cfg), D, 10.0, alphas, torch.device("cpu"), (1, D), None)
    load_synth(ddpm.model, 31, "trained")
    rs = np.random.RandomState(401)
    cond = torch.from_numpy(rs.uniform(0, 1, (B, C)).astype(np.float32))
    y_T, z = replay_sample_noise(seed, B, D, T)
    out = dict(cond=cond.numpy(), y_T=y_T.numpy(), T=np.int64(T), z=np.stack([z[i].numpy() for i in range(T - 1, 1, -1)]))
    with torch.no_grad():
        for omega in (0.0, 1.0):
            torch.manual_seed(seed)
            out[f"om{omega:g}_y0"] = ddpm.sample(cond, omega).numpy()
        d64 = RMSR.DDPM(T, ref_unet(cfg), D, 10.0, alphas, torch.device("cpu"), (1, D), None)
        load_synth(d64.model, 31, "trained")
        d64 = d64.double()
        for omega in (0.0, 1.0):
            y = y_T.double()
            c64 = cond.double()
            for i in range(T - 1, -1, -1):
                t = (torch.full((1, B), i) / T).double()
                e0 = d64.model(y, t, c64, torch.zeros(B, 1, dtype=torch.float64))
                e1 = d64.model(y, t, c64, torch.ones(B, 1, dtype=torch.float64))
                nz = z[i].double() if i > 1 else 0
                e = (1 + omega) * e1 - omega * e0
                y = (y - d64.betas[i] / d64.sqrt_one_minus_alphas_cumprod[i] * e) * d64.reciprocal_sqrt_alphas[i] \
                    + (1.0 - d64.alphas_cumprod[i - 1 if i - 1 >= 0 else 0]) / (1.0 - d64.alphas_cumprod[i]) * nz
                if i > T - 5:
                    y = (y - torch.mean(y)) / torch.sqrt(torch.var(y))
            out[f"om{omega:g}_y0_f64"] = y.numpy()
            print("T=1000 omega", omega, "ref f32 vs f64 rel err",
                  float(np.abs(out[f"om{omega:g}_y0"] - y.numpy()).max() / np.abs(y.numpy()).max()))
    save("g4_sample_msr3_T1000.npz", **out)


# ---------------------------------------------------------------- G11
def g11():
    """utils/dataset_generate.py:147-245: CONV_CO_MINLP_GEN (exhaustive-search labels of the CO problem), float64.
    The reference calls `np.alltrue`, which numpy 2 removed: it is aliased to `np.all` (its definition) for this run only."""
    import contextlib, io
    import utils.dataset_generate as DG
    had = hasattr(np, "alltrue")
    if not had:
        np.alltrue = np.all
    out = {}
    for tag, n, samples, seed in (("n2", 2, 6, 2024), ("n3", 3, 8, 2025), ("n4", 4, 1, 2026)):
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()) as buf, contextlib.redirect_stderr(io.StringIO()):
            X, Y = DG.CONV_CO_MINLP_GEN(n, samples)
        out[tag + "_X"], out[tag + "_Y"], out[tag + "_seed"] = X, Y, np.int64(seed)
        print(tag, X.shape, Y.shape, buf.getvalue().splitlines()[0])
    if not had:
        del np.alltrue
    save("g11_co_minlp.npz", **out)


if __name__ == "__main__":
    groups = dict(G1=g1, G2=g2, G3=g3, G4=g4, G5=g5, G6=g6, G7=g7, G8=g8, G9=g9, G10=g10, G11=g11)
    for g in (sys.argv[1:] or list(groups)):
        groups[g]()
