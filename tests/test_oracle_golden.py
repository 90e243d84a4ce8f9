"""Pin the CPU oracle against vectors captured from the imported reference.

The oracle issues the reference's ATen ops in the reference's order, so float32 results are expected to be
bit-identical on the torch build that generated the goldens; the asserted tolerance is 2e-6 relative so that a
different CPU / torch build (different GEMM blocking) still passes.
"""
import json
import os

import numpy as np
import pytest
import torch

from _util import GOLD, synth_params
from oracle import ddpm_oracle as O
from weights import CONFIGS

RTOL = 2e-6


def close(a, b, rtol=RTOL, atol=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, f"max rel-to-max err {err:.3e} > {rtol}"
    return err


@pytest.mark.parametrize("T", [4, 20, 400, 500, 1000])
def test_g1_schedule(gold, T):
    g = gold("g1_schedule.npz")
    betas = O.cosine_betas(T)
    assert np.array_equal(betas, g[f"T{T}_betas_f64"])  # float64 host arithmetic: exact
    bufs = O.schedule_buffers(1.0 - betas)
    assert list(bufs) == list(O.BUFFER_NAMES)
    for k, v in bufs.items():
        assert np.array_equal(v.numpy(), g[f"T{T}_{k}"]), k


@pytest.mark.parametrize("name,flavour,seed", [
    ("msr3", "trained", 11), ("msr80", "trained", 11), ("msr80", "init", 12), ("co3", "trained", 11),
    ("nu3", "trained", 11), ("tiny", "trained", 11), ("tiny", "init", 12)])
def test_g2_unet_forward(gold, name, flavour, seed):
    g = gold(f"g2_unet_{name}_{flavour}.npz")
    plan, p = synth_params(name, seed, flavour)
    x, cond = torch.from_numpy(g["x"]), torch.from_numpy(g["cond"])
    B = x.shape[0]
    with torch.no_grad():
        taps = {}
        ts = torch.from_numpy(g["a_ts"])
        eps = O.unet_forward(p, plan, x, ts / int(g["a_T"]), cond, torch.from_numpy(g["a_mask"]), taps)
        close(eps, g["a_eps"])
        for k in g.files:
            if k.startswith("a_tap."):
                close(taps[k[len("a_tap."):]], g[k])
        t = torch.full((1, B), int(g["b_step"]), dtype=torch.int64) / 20
        close(O.unet_forward(p, plan, x, t, cond, torch.zeros(B, 1)), g["b_eps"])
        close(O.unet_forward(p, plan, x, t, cond, torch.ones(B, 1)), g["c_eps"])


@pytest.mark.parametrize("name", ["tiny", "nu3", "msr80"])
def test_g3_loss_and_grads(gold, name):
    g = gold(f"g3_loss_{name}.npz")
    plan, p = synth_params(name, 21)
    T = int(g["T"])
    bufs = O.schedule_buffers(1.0 - O.cosine_betas(T))
    loss, grads = O.ddpm_loss_and_grads(p, plan, bufs, T, torch.from_numpy(g["y"]), torch.from_numpy(g["cond"]),
                                        torch.from_numpy(g["ts"]), torch.from_numpy(g["noise"]),
                                        torch.from_numpy(g["mask"]))
    close(loss, g["loss"])
    gmax = max(float(v.abs().max()) for v in grads.values())
    for k, v in grads.items():
        if name == "tiny":
            ref = g["grad." + k]
            assert np.abs(v.numpy() - ref).max() <= 2e-6 * gmax + 1e-6 * np.abs(ref).max(), k
        else:
            nrm = float(torch.sqrt((v.double() ** 2).sum()))
            assert abs(nrm - float(g["gradnorm." + k])) <= 1e-5 * max(float(g["gradnorm." + k]), 1e-3 * gmax), k
            ref = g["gradhead." + k]
            assert np.abs(v.reshape(-1)[:16].numpy() - ref).max() <= 5e-6 * gmax, k


def _z_dict(z, T):
    return {i: torch.from_numpy(z[j]) for j, i in enumerate(range(T - 1, 1, -1))}


def test_g4_sample_nu_checkpoint(gold):
    g = gold("g4_sample_nu_ckpt.npz")
    cfg = CONFIGS["nu3"]
    plan = O.unet_plan(cfg["input_dim"], cfg["proj_dim"], cfg["cond_dim"], cfg["dims"], cfg["n_blocks"])
    p = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    assert list(p) == list(O.state_shapes(plan))
    T = int(g["T"])
    bufs = O.schedule_buffers(1.0 - O.cosine_betas(T))
    cond, y_T, z = torch.from_numpy(g["cond"]), torch.from_numpy(g["y_T"]), _z_dict(g["z"], T)
    for omega, tol in ((0.0, RTOL), (1.0, RTOL), (500.0, 1e-4)):
        trace = []
        y0 = O.ddpm_sample(p, plan, bufs, T, cond, omega, y_T, z, trace)
        close(y0, g[f"om{omega:g}_y0"], rtol=tol)
        if omega == 1.0:
            close(np.stack([e.numpy() for _, _, e in trace]), g["om1_eps_steps"], rtol=1e-5)
    # known answer: less ratio 0.91359 on these rows at omega = 500
    y0 = torch.from_numpy(g["om500_y0"])
    Xs = cond.clone()
    Xs[:, 0::2] *= 400
    Xs[:, 1::2] *= 400
    Yd = O.nu_decode(y0, 400, 400, float(g["P_sum"]))
    Yt = torch.from_numpy(g["y_test"]).clone()
    Yt[:, 0] *= 400; Yt[:, 1] *= 400; Yt[:, 2:] *= float(g["P_sum"])
    pr, tr = O.nu_rate(Yd, Xs), O.nu_rate(Yt, Xs)
    close(pr, g["om500_pred_rate"], rtol=1e-5)
    ratio = float(pr.sum() / tr.sum())
    assert abs(ratio - 0.91359) < 2e-5 and abs(ratio - float(g["om500_less_ratio"])) < 1e-6
    # error budget: float32 vs float64 evaluation of the same trajectory (hard parts)
    for omega, lim in ((0.0, 1e-5), (1.0, 1e-5), (500.0, 2e-2)):
        f64 = g[f"om{omega:g}_y0_f64"]
        rel = np.abs(g[f"om{omega:g}_y0"] - f64).max() / np.abs(f64).max()
        assert rel < lim, (omega, rel)
    # the float64 path of the oracle reproduces the reference's float64 trajectory
    p64 = {k: v.double() for k, v in p.items()}
    b64 = O.schedule_buffers(1.0 - O.cosine_betas(T))  # buffers stay float32-valued, as registered
    b64 = {k: v.double() for k, v in b64.items()}
    y64 = O.ddpm_sample(p64, plan, b64, T, cond.double(), 500.0, y_T.double(), {i: v.double() for i, v in z.items()})
    close(y64, g["om500_y0_f64"], rtol=1e-9)


@pytest.mark.parametrize("name,T", [("tiny", 8), ("msr80", 6), ("msr3", 6), ("co3", 6), ("tiny", 3)])
def test_g4_sample_synth(gold, name, T):
    g = gold(f"g4_sample_{name}_T{T}.npz")
    plan, p = synth_params(name, 31)
    bufs = O.schedule_buffers(1.0 - O.cosine_betas(T))
    z = _z_dict(g["z"], T)
    for omega in (0.0, 1.0, 3.0):
        y0 = O.ddpm_sample(p, plan, bufs, T, torch.from_numpy(g["cond"]), omega, torch.from_numpy(g["y_T"]), z)
        close(y0, g[f"om{omega:g}_y0"], rtol=1e-5)


def test_g4_sample_T1000(gold):
    """BASELINE config 2's schedule length (G10): 1 000 steps = 2 000 chained forwards, 16 rows."""
    g = gold("g4_sample_msr3_T1000.npz")
    T = int(g["T"])
    plan, p = synth_params("msr3", 31)
    bufs = O.schedule_buffers(1.0 - O.cosine_betas(T))
    z = _z_dict(g["z"], T)
    y0 = O.ddpm_sample(p, plan, bufs, T, torch.from_numpy(g["cond"]), 1.0, torch.from_numpy(g["y_T"]), z)
    close(y0, g["om1_y0"], rtol=1e-5)
    assert close(g["om1_y0"], g["om1_y0_f64"], rtol=2e-5) > 0     # the reference's own float32 error over 1 000 steps


def test_g5_decoders(gold):
    g = gold("g5_decoders.npz")
    t = lambda k: torch.from_numpy(g[k])
    dec = O.msr_decode(t("msr_y"))
    close(dec, g["msr_dec"])
    close(O.msr_rate(10.0 * dec, t("msr_gain")), g["msr_rate"])
    dec = O.co_decode(t("co_y"))
    close(dec, g["co_dec"])
    assert float(dec[5].abs().sum()) == 0.0
    close(O.co_cost(t("co_X"), dec), g["co_cost"])
    dec = O.nu_decode(t("nu_y"), 400, 400, 18.0)
    close(dec, g["nu_dec"])
    close(O.nu_rate(dec, t("nu_X")), g["nu_rate"], rtol=1e-5)


def test_g7_state_layout_and_ema(gold):
    with open(os.path.join(GOLD, "g7_state_layout.json")) as f:
        layout = json.load(f)
    for name, cfg in CONFIGS.items():
        plan = O.unet_plan(cfg["input_dim"], cfg["proj_dim"], cfg["cond_dim"], cfg["dims"], cfg["n_blocks"])
        shapes = O.state_shapes(plan)
        ref = [(k, tuple(s)) for k, s, _ in layout[name]]
        model_keys = [(k[len("model."):], s) for k, s in ref if k.startswith("model.")]
        assert model_keys == [(k, tuple(s)) for k, s in shapes.items()], name
        assert [k for k, _ in ref[:8]] == list(O.BUFFER_NAMES)
        ema_keys = [k[len("ema.module."):] for k, _ in ref if k.startswith("ema.module.")]
        assert ema_keys == list(shapes)
        assert ref[8 + len(shapes)][0] == "ema.n_averaged"
    g = gold("g7_ema.npz")
    plan, cur = synth_params("tiny", 41)
    avg, n = {k: torch.zeros_like(v) for k, v in cur.items()}, 0
    for step in range(3):
        cur = {k: v + 0.01 * (step + 1) for k, v in cur.items()}
        avg, n = O.ema_update(avg, cur, 0.9, n)
        assert n == int(g[f"step{step}_n"])
        close(avg["feature_proj.weight"], g[f"step{step}_feature_proj.weight"])
        close(avg["norm.bias"], g[f"step{step}_norm.bias"])


def test_g8_sum_rate_label_generator(gold):
    """oracle/sumrate_oracle.py against the reference's SUM_RATE_GEN / alpha_calc outputs (float64, same additions)."""
    from oracle import sumrate_oracle as S
    g = gold("g8_sum_rate_gen.npz")
    assert np.array_equal(S.sum_rate_grad(g["step_gs"], g["step_schemes"]), g["step_grad"])
    assert np.allclose(S.alpha_calc(g["step_grad"]), g["step_alpha"], rtol=1e-13, atol=1e-15)
    for tag in ("m3", "m7", "m80"):
        rates, schemes = S.sum_rate_gen(g[tag + "_gs"], float(g[tag + "_W"]))
        assert np.allclose(schemes, g[tag + "_schemes"], rtol=1e-11, atol=1e-13), tag
        assert np.allclose(rates, g[tag + "_rates"], rtol=1e-12), tag
        assert np.allclose(schemes.sum(1), float(g[tag + "_W"]), rtol=1e-12), tag     # total power is conserved


@pytest.mark.parametrize("tag,n", [("n2", 2), ("n3", 3), ("n4", 4)])
def test_g11_co_minlp_label_generator(gold, tag, n):
    """CONV_CO_MINLP_GEN (utils/dataset_generate.py:147-245): the restatement replays the reference's numpy draws from the
    same seed and must reproduce its features and labels (decision | allocation | cost) bit for bit."""
    from oracle import co_minlp_oracle as C
    g = gold("g11_co_minlp.npz")
    Xr, Yr = g[tag + "_X"], g[tag + "_Y"]
    np.random.seed(int(g[tag + "_seed"]))
    X, Y, hits = C.conv_co_minlp_gen(n, Xr.shape[0])
    assert np.array_equal(X, Xr)
    assert np.array_equal(Y, Yr), np.abs(Y - Yr).max()
    # domain properties: an offloading decision's allocation sums to 1 and is zero where nothing is offloaded
    D, F = Y[:, :n], Y[:, n:2 * n]
    assert np.all((F > 0) == (D > 0)) and np.allclose(F.sum(1)[D.sum(1) > 0], 1.0, atol=1e-5)
