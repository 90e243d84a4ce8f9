"""CPU tests of the host side: the C-ABI library loads and exports what include/diffsg.h declares, the module tree has
the reference's state-dict layout and seeded init, loaders / decoders / schedule match the reference-generated goldens,
and the compute entry points fail loudly without a GPU (no fallback).  No compute call is made here."""
import json
import os
import re

import numpy as np
import pytest
import torch

from _util import GOLD, ROOT
from weights import CONFIGS


def test_library_exports_every_declared_symbol():
    from diffsg_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "diffsg.h")).read()
    declared = sorted(set(re.findall(r"\b(dsg_[a-z_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in diffsg.h but not exported"
    assert sorted(_lib.exported_symbols()) == declared


@pytest.mark.parametrize("name", list(CONFIGS))
def test_state_dict_layout(name):
    from diffsg_amd import UNet1D, generate_cosine_schedule
    from diffsg_amd.classifier_free_MSR import DDPM
    with open(os.path.join(GOLD, "g7_state_layout.json")) as f:
        ref = json.load(f)[name]
    cfg = CONFIGS[name]
    D = cfg["input_dim"]
    m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
    d = DDPM(20, m, D, 10.0, 1.0 - generate_cosine_schedule(20), torch.device("cpu"), (1, D), None)
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in d.state_dict().items()]
    assert got == ref


@pytest.mark.parametrize("name", ["tiny", "nu3"])
def test_seeded_construction_matches_reference(gold, name):
    from diffsg_amd import UNet1D, init_weights
    g = gold("g7_seeded_init.npz")
    cfg = CONFIGS[name]
    torch.manual_seed(5)
    m = UNet1D(**cfg, is_attn=(False,) * len(cfg["dims"]))
    m.apply(init_weights)
    sums = np.array([float(v.double().sum()) for v in m.state_dict().values()])
    absum = np.array([float(v.double().abs().sum()) for v in m.state_dict().values()])
    assert np.array_equal(sums, g[name + "_sums"]) and np.array_equal(absum, g[name + "_abs"])


def test_nu_checkpoint_weights_load_strict(gold):
    from diffsg_amd import UNet1D
    g = gold("g4_sample_nu_ckpt.npz")
    cfg = CONFIGS["nu3"]
    m = UNet1D(**cfg, is_attn=(False,) * 3)
    m.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}, strict=True)


@pytest.mark.parametrize("T", [4, 20, 400, 500, 1000])
def test_schedule_and_buffers(gold, T):
    from diffsg_amd import generate_cosine_schedule
    from diffsg_amd.classifier_free_MSR import DDPM
    g = gold("g1_schedule.npz")
    betas = generate_cosine_schedule(T)
    assert np.array_equal(betas, g[f"T{T}_betas_f64"])
    d = DDPM(T, torch.nn.Identity(), 3, 10.0, 1.0 - betas, "cpu", (1, 3))
    names = [k for k in d.state_dict() if not k.startswith("ema.")]
    assert names == ["betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                     "reciprocal_sqrt_alphas", "remove_noise_coeff", "sqrt_betas"]
    for k in names:
        assert np.array_equal(d.state_dict()[k].numpy(), g[f"T{T}_{k}"]), k
    # per-step coefficient table against the reference's scalar expressions (MSR.py:133-134)
    coef = d._coef_table()
    for i in (0, 1, 2, T - 1):
        c1 = d.betas[i] / d.sqrt_one_minus_alphas_cumprod[i]
        c3 = (1.0 - d.alphas_cumprod[i - 1 if i - 1 >= 0 else 0]) / (1.0 - d.alphas_cumprod[i])
        assert coef[i, 0] == c1 and coef[i, 1] == d.reciprocal_sqrt_alphas[i] and coef[i, 2] == c3
        assert float(coef[i, 3]) == (1.0 if i > 1 else 0.0)


def test_loaders_match_reference(gold):
    from diffsg_amd.classifier_free_CO import co_data_load
    from diffsg_amd.classifier_free_MSR import msr_data_load
    from diffsg_amd.classifier_free_NU import nu_data_load
    g = gold("g6_loaders.npz")
    dd = os.path.join(GOLD, "data")
    Xtr, Ytr, Xte, Yte, cfg = msr_data_load(os.path.join(dd, "3c_10w_200samples.csv"))
    for a, k in ((Xtr, "msr_Xtr"), (Ytr, "msr_Ytr"), (Xte, "msr_Xte"), (Yte, "msr_Yte")):
        assert np.array_equal(a, g[k]), k
    assert [cfg["M"], cfg["W"], cfg["scaler_min"], cfg["scaler_max"]] == list(g["msr_cfg"])
    assert (cfg["sfn"], cfg["cfn"], cfg["cdim"]) == (1, 0, 1)
    Xtr, Ytr, Xte, Yte, Rte, cfg = nu_data_load(os.path.join(dd, "3u_18mW_200samples.csv"), 400, 400)
    for a, k in ((Xtr, "nu_Xtr"), (Ytr, "nu_Ytr"), (Xte, "nu_Xte"), (Yte, "nu_Yte"), (Rte, "nu_Rte")):
        assert np.array_equal(a, g[k]), k
    assert [cfg["K"], cfg["P_sum"]] == list(g["nu_cfg"]) and (cfg["width"], cfg["height"]) == (400, 400)
    Xtr, Ytr, Xte, Yte, cfg = co_data_load(os.path.join(dd, "3nodes_200samples_ood.csv"))
    for a, k in ((Xtr, "co_Xtr"), (Ytr, "co_Ytr"), (Xte, "co_Xte"), (Yte, "co_Yte")):
        assert a.shape == g[k].shape and np.allclose(a, g[k], rtol=1e-13, atol=0), k
    assert np.allclose([cfg["scaler_min"], cfg["scaler_max"]], g["co_cfg"], rtol=1e-13)


def test_decoders_have_no_cpu_path():
    """The decoders / evaluators are device kernels behind the C ABI (tests/test_gpu_parity.py pins them to the goldens);
    CPU tensors are refused instead of silently taking another path."""
    from diffsg_amd import decode as Dc
    y = torch.rand(4, 5)
    for fn, args in ((Dc.msr_decode, (y,)), (Dc.co_decode, (y,)), (Dc.row_softmax, (y,)), (Dc.msr_rate, (y, y)),
                     (Dc.co_cost, (torch.rand(4, 15), y)), (Dc.nu_decode, (y, 400, 400, 18.0)), (Dc.nu_rate, (y, torch.rand(4, 6)))):
        with pytest.raises(RuntimeError, match="no CPU path"):
            fn(*args)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_compute_entry_points_fail_loudly_without_gpu():
    from diffsg_amd import UNet1D, generate_cosine_schedule
    from diffsg_amd.classifier_free_MSR import DDPM, train_ddpm_msr
    cfg = CONFIGS["tiny"]
    m = UNet1D(**cfg, is_attn=(False,) * 3)
    x, c = torch.zeros(4, 3), torch.zeros(4, 3)
    with pytest.raises(RuntimeError):
        m(x, torch.zeros(1, 4), c, torch.ones(4, 1))
    d = DDPM(20, m, 3, 10.0, 1.0 - generate_cosine_schedule(20), torch.device("cpu"), (1, 3), None)
    with pytest.raises(RuntimeError):
        d.sample(c, 1.0)
    with pytest.raises(RuntimeError):
        d(x, c)
    with pytest.raises(RuntimeError):
        train_ddpm_msr(os.path.join(GOLD, "data", "3c_10w_200samples.csv"), epochs=1)


def test_flatten_parameters_keeps_state_dict_and_aliases_one_buffer():
    """train.flatten_parameters: same keys, shapes and values; every parameter is a slice of the flat tensor in
    state-dict order (the order of the gradient bucket)."""
    from diffsg_amd import UNet1D
    from diffsg_amd.train import flatten_parameters
    torch.manual_seed(0)
    net = UNet1D(3, 16, 6, dims=(8, 4), is_attn=(False, False), n_blocks=1)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    flat = flatten_parameters(net)
    after = net.state_dict()
    assert list(after) == list(before) and flat.numel() == sum(v.numel() for v in before.values())
    off = 0
    for k, v in after.items():
        assert torch.equal(v, before[k]), k
        assert v.data_ptr() == flat.data_ptr() + 4 * off, k
        off += v.numel()
    flat.mul_(2.0)
    assert torch.equal(net.state_dict()[list(before)[0]], 2.0 * before[list(before)[0]])


def test_co_validation_data_gen_matches_reference(gold):
    """classifier_free_CO.py:416-449: same numpy draws in the same order, same split."""
    from diffsg_amd import classifier_free_CO as CO
    g = gold("g9_co_validation.npz")
    np.random.seed(321)
    Xtr, Ytr, Xte, Yte, cfg = CO.validation_data_gen()
    assert [list(a.shape) for a in (Xtr, Ytr, Xte, Yte)] == g["shapes"].tolist()
    assert cfg == {'sfn': int(g["sfn"]), 'cfn': int(g["cfn"])}
    np.testing.assert_array_equal(Xtr[:16], g["Xtr_head"]); np.testing.assert_array_equal(Ytr[:16], g["Ytr_head"])
    np.testing.assert_array_equal(Xte[-16:], g["Xte_tail"]); np.testing.assert_array_equal(Yte[-16:], g["Yte_tail"])
    np.testing.assert_allclose([Xtr.sum(), Ytr.sum(), Xte.sum(), Yte.sum()], g["sums"], rtol=1e-13)


def test_trajectory_writers_fail_loudly_without_gpu(tmp_path):
    from diffsg_amd import trajectory
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        trajectory.nu_trajectory_gen_store(None, os.path.join(GOLD, "data", "3u_18mW_200samples.csv"), str(tmp_path / "t.csv"))


# Kernels a 65 536-row reverse step launches (launch table) and the training step's large kernels.
# none of them may use scratch memory (a spilled register is a vector-memory round trip inside the inner loop).
SAMPLING_STEP_KERNELS = [
    "dsg::k_linear_h<4, 1, 0, false>",          # feature_proj
    "dsg::k_panel128_h<false, 0, 1, 2>", "dsg::k_panel128_h<false, 1, 2, 2>", "dsg::k_panel128_h<true, 0, 1, 2>", "dsg::k_panel128_h<true, 2, 3, 2>",
    "dsg::k_panel128_h<false, 0, 1, 4>", "dsg::k_panel128_h<false, 1, 2, 4>", "dsg::k_panel128_h<true, 0, 1, 4>", "dsg::k_panel128_h<true, 2, 3, 4>",
    "dsg::k_res64_dual", "dsg::k_res64_lds<true, 0>", "dsg::k_res64_lds<true, 4>",
    "dsg::k_fused_narrow_lds<2>", "dsg::k_fused_narrow_lds<3>", "dsg::k_fused_narrow_lds<0>",
    "dsg::k_update", "dsg::k_renorm_sum", "dsg::k_renorm_apply",
]
OTHER_HOT_KERNELS = [
    "dsg::k_res64_lds<false, 2>", "dsg::k_res64_lds<false, 1>", "dsg::k_res64_lds<true, 2>", "dsg::k_res64_lds<false, 0>",
    "dsg::k_fused_narrow_h<true, 0>", "dsg::k_fused_narrow_h<false, 0>", "dsg::k_fused_narrow_h<true, 2>", "dsg::k_fused_narrow_h<false, 2>",
    "dsg::k_fused_narrow_h<true, 3>", "dsg::k_fused_narrow_h<false, 3>",
    "dsg::k_wgrad_h", "dsg::k_fused_narrow_bwd_h", "dsg::k_resblock_bwd_c<128, true>", "dsg::k_resblock_bwd_c<128, false>",
    "dsg::k_resblock_bwd_c<64, true>", "dsg::k_resblock_bwd_c<64, false>", "dsg::k_wide128_h<true, 0, 1>", "dsg::k_wide128_h<false, 0, 1>",
    "dsg::k_resblock_h<64, true>", "dsg::k_resblock_h<64, false>", "dsg::k_cond_embed_h", "dsg::k_colsum",
    # every other instantiated >= 64-wide kernel of the training step and of the small-launch forms
    "dsg::k_resblock_c<128, true>", "dsg::k_resblock_c<128, false>", "dsg::k_resblock_c<64, true>", "dsg::k_resblock_c<64, false>",
    "dsg::k_resblock_h<128, true>", "dsg::k_resblock_h<128, false>", "dsg::k_wide128_h<false, 1, 2>", "dsg::k_wide128_h<true, 2, 3>",
    "dsg::k_linear_h<4, 0, 0, false>", "dsg::k_linear_h<2, 0, 0, false>", "dsg::k_linear_h<3, 0, 1, true>",
]
# the exact-float32 reverse step at the bench size: the unrolled-chain forms of the wide blocks and the plain Linears
F32_STEP_KERNELS = [
    "dsg::k_resblock<128, true, true>", "dsg::k_resblock<128, false, true>", "dsg::k_resblock<64, true, true>", "dsg::k_resblock<64, false, true>",
    "dsg::k_linear<4, 1, 0, false>", "dsg::k_linear<3, 0, 1, true>", "dsg::k_linear<2, 0, 0, false>", "dsg::k_linear<4, 0, 0, false>",
    "dsg::k_resblock_lin<128, false, 2, false>", "dsg::k_resblock_lin<128, true, 3, true>",      # the pairs of the bench shape
]
# kernels that MAY keep a few bytes of scratch (stated, bounded): the whole-net tile kernel is the union of every small-launch body under
# one 256-register budget (two workgroups per CU); what it spills is reloaded once per operator, not inside a loop
BOUNDED_SCRATCH_KERNELS = {"dsg::k_unet_tile<2>": 32, "dsg::k_unet_tile<3>": 32, "dsg::k_unet_tile<0>": 32,
                           "dsg::k_fused_narrow": 36,
                           "dsg::k_resblock_lin<64, true, 4, false>": 20}      # up.14.res + up.15.lin: four registers stored / reloaded once per tile      # the exact path's narrow run: 8 spilled registers under its 128-register bound


TABLE_DRIVEN_KERNELS = ["k_fused_narrow_lds", "k_fused_narrow_h", "k_fused_narrow_bwd_h", "k_wgrad_h", "k_wgrad", "k_colsum", "k_fused_narrow"]


def test_table_driven_kernels_issue_no_flat_memory_instructions(tmp_path):
    """Kernels that read their operands' addresses from descriptor tables in memory (the fused narrow run forward / backward, the
    weight-gradient and column-sum launches) declare them global (csrc/dsg_kernels.hpp, as_global): a pointer loaded from memory is
    generic to hipcc, every access through it a FLAT instruction, and every wait for an LDS read then also waits for all global loads in
    flight.  Disassembles the gfx950 code object of the built library: no flat_load / flat_store / flat_atomic in those kernels."""
    import re, shutil, subprocess
    from diffsg_amd import _lib
    _lib.build()
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    cos = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(cos) == 1, os.listdir(tmp_path)
    dis = subprocess.run([objdump, "-d", str(tmp_path / cos[0])], check=True, capture_output=True, text=True).stdout
    syms = [(m.start(), m.group(1)) for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:$", dis, re.M)]
    seen, bad = set(), {}
    for k, (pos, name) in enumerate(syms):
        m = re.match(r"_ZN3dsg\d+([A-Za-z_0-9]+?)(?:I|E)", name)
        if not m or m.group(1) not in TABLE_DRIVEN_KERNELS:
            continue
        body = dis[pos:syms[k + 1][0] if k + 1 < len(syms) else len(dis)]
        seen.add(m.group(1))
        n_flat = len(re.findall(r"\bflat_(?:load|store|atomic)", body))
        assert len(re.findall(r"\bglobal_(?:load|store|atomic)", body)) > 0, name
        if n_flat:
            bad[name] = n_flat
    assert seen == set(TABLE_DRIVEN_KERNELS), sorted(set(TABLE_DRIVEN_KERNELS) - seen)
    assert not bad, bad


def test_no_partial_register_write_hazard_behind_the_fp16_split():
    """`split_pair_nowait` (csrc/dsg_split.hpp) drops the wait state behind v_fma_mixhi_f16, which writes HALF a
    register; on gfx940+ a vector / matrix instruction issued directly behind it that names that register reads the OLD contents, and
    hipcc does not pad inline asm.  The call sites place something else there by construction, but the asm is not pinned: this check
    disassembles every kernel of the built library and requires that no v_fma_mixhi_f16 is directly followed by a v_* instruction that
    mentions its destination (tools/isa_lint.py, mixhi_hazards).  Runs on the CPU on every build."""
    import importlib.util
    from diffsg_amd import _lib
    _lib.build()
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not found")
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    # the checker itself: a dependent pair is caught, a separated pair and an independent neighbour are not
    assert lint.mixhi_hazards(["v_fma_mixhi_f16 v5, v1, 1.0, -v2 op_sel:[0,0,1]", "v_pk_mul_f32 v[8:9], v[4:5], v[6:7]"])
    assert lint.mixhi_hazards(["v_fma_mixhi_f16 v5, v1, 1.0, -v2", "v_mfma_f32_32x32x16_f16 a[0:15], v[2:5], v[6:9], a[0:15]"])
    assert not lint.mixhi_hazards(["v_fma_mixhi_f16 v5, v1, 1.0, -v2", "s_nop 0", "v_mov_b32_e32 v6, v5"])
    assert not lint.mixhi_hazards(["v_fma_mixhi_f16 v5, v1, 1.0, -v2", "v_mov_b32_e32 v6, v15"])
    dis = lint.disassemble(_lib.LIB_PATH)
    n_mixhi, bad = 0, {}
    for name, ins in lint.kernels(dis):
        n_mixhi += sum(1 for t in ins if t.startswith("v_fma_mixhi_f16"))
        hz = lint.mixhi_hazards(ins)
        if hz:
            bad[name] = [(a, b) for _, a, b in hz[:3]]
    assert n_mixhi > 1000, n_mixhi            # the split is in every split-path kernel: the disassembly was read
    assert not bad, bad


@pytest.mark.parametrize("group", ["sampling_step", "other_hot", "f32_step"])
def test_hot_kernels_use_no_scratch_memory(group):
    """Compiled with -Rpass-analysis=kernel-resource-usage (diffsg_amd/_lib.build keeps hipcc's remarks beside the library): every
    kernel of the bench-size reverse step, and the large kernels of the training step / the other launch forms, report
    ScratchSize = 0 bytes per lane.  Runs on the CPU: hipcc cross-compiles."""
    from diffsg_amd import _lib
    _lib.build()
    res = _lib.kernel_resources()
    names = {"sampling_step": SAMPLING_STEP_KERNELS, "other_hot": OTHER_HOT_KERNELS, "f32_step": F32_STEP_KERNELS}[group]
    missing = [n for n in names if n not in res]
    assert not missing, f"kernels not in the build record (renamed?): {missing}"
    spilled = {n: res[n]["scratch"] for n in names if res[n]["scratch"] != 0}
    assert not spilled, f"scratch bytes per lane: {spilled}"
    if group == "other_hot":
        assert not [n for n in res if "k_resblock_bwd_h<128" in n or "k_resblock_bwd_h<64" in n], "the one-wave-per-tile wide backward is gone"
        over = {n: res[n]["scratch"] for n, cap in BOUNDED_SCRATCH_KERNELS.items() if res[n]["scratch"] > cap}
        assert not over, f"scratch above the stated bound: {over}"
