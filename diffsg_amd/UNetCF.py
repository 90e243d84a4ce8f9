"""UNet1D -- the conditional MLP "1-D U-Net" denoiser (reference: ddpm_opt/UNetCF.py:260-356).

The module tree below exists only to own the parameters under the reference's state-dict keys (SURVEY.md 5.4:
`feature_proj`, `time_emb.lin{1,2}`, `down.<i>.res.*` / `down.<i>.lin`, `middle.res{1,2}`, `up.<i>.res.*` /
`up.<i>.lin`, `norm`, `final`) and to construct them in the reference's order, so a seeded construction gives the
reference's initial weights and `load_state_dict(strict=True)` accepts the reference's checkpoints.  No arithmetic
happens in Python: `forward` hands device pointers to libdiffsg_hip.so.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _lib


class _ParamsOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("parameter holder: the arithmetic lives in libdiffsg_hip.so (call UNet1D.forward)")


class TimeEmbedding(_ParamsOnly):
    """UNetCF.py:17-28."""

    def __init__(self, in_dim):
        super().__init__()
        self.in_dim = in_dim
        self.lin1 = nn.Linear(in_dim // 4, in_dim)
        self.lin2 = nn.Linear(in_dim, in_dim)


class ResidualBlock(_ParamsOnly):
    """UNetCF.py:49-81 (registration order matters for RNG parity and state-dict order)."""

    def __init__(self, in_dim, out_dim, time_dim, cond_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(in_dim)
        self.lin1 = nn.Linear(in_dim, out_dim)
        self.norm2 = nn.LayerNorm(out_dim)
        self.lin2 = nn.Linear(out_dim, out_dim)
        self.norm3 = nn.LayerNorm(out_dim)
        self.lin3 = nn.Linear(out_dim, out_dim)
        if in_dim != out_dim:
            self.shortcut = nn.Linear(in_dim, out_dim)
        self.time_emb = nn.Linear(time_dim, out_dim)
        self.cond_emb = nn.Linear(cond_dim, out_dim)


class _ResHolder(_ParamsOnly):
    def __init__(self, in_dim, out_dim, time_dim, cond_dim):
        super().__init__()
        self.res = ResidualBlock(in_dim, out_dim, time_dim, cond_dim)


class DownBlock(_ResHolder):
    """UNetCF.py:160-168."""


class UpBlock(_ResHolder):
    """UNetCF.py:182-192: the block sees cat(x, skip), i.e. in_dim + out_dim features."""

    def __init__(self, in_dim, out_dim, time_dim, cond_dim):
        super().__init__(in_dim + out_dim, out_dim, time_dim, cond_dim)


class MiddleBlock(_ParamsOnly):
    """UNetCF.py:206-215."""

    def __init__(self, dim, time_dim, cond_dim):
        super().__init__()
        self.res1 = ResidualBlock(dim, dim, time_dim, cond_dim)
        self.res2 = ResidualBlock(dim, dim, time_dim, cond_dim)


class _LinHolder(_ParamsOnly):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.lin = nn.Linear(in_dim, out_dim)


class Upsample(_LinHolder):
    """UNetCF.py:230-233."""


class Downsample(_LinHolder):
    """UNetCF.py:245-248."""


# In-place parameter updates that PyTorch's tensor version counters do not see (fused optimizers -- torch._fused_adam_ leaves
# `_version` untouched -- and anything that writes through raw pointers) would leave the library's packed weights stale.
# A step of an optimizer that OWNS parameters of a UNet1D (its own tensors, or storage they alias: train.FlatAdam) bumps
# that model's epoch, and the model re-packs on its next call; optimizers over unrelated parameters leave it alone.
import weakref  # noqa: E402

_NATIVES = weakref.WeakSet()
_OPT_OWNERS = weakref.WeakKeyDictionary()   # optimizer -> (signature of its param groups, [weakref to _Native])


def _on_optimizer_step(optimizer, *_args, **_kw):
    groups = optimizer.param_groups
    sig = tuple(len(g["params"]) for g in groups) + (len(_NATIVES),)
    hit = _OPT_OWNERS.get(optimizer)
    if hit is None or hit[0] != sig:
        ptrs = set()
        for g in groups:
            for p in g["params"]:
                st = p.untyped_storage()
                ptrs.add((st.data_ptr(), p.device))
        owners = [weakref.ref(n) for n in list(_NATIVES) if n.storage_keys & ptrs]
        hit = (sig, owners)
        _OPT_OWNERS[optimizer] = hit
    for r in hit[1]:
        n = r()
        if n is not None:
            n.epoch += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook  # noqa: E402

_reg_post_hook(_on_optimizer_step)


class _Native:
    """Owner of the dsg_handle; never copied or pickled with the module."""

    def __init__(self):
        self.handle = None
        self.bound_key = None
        self.epoch = 0
        self.named_params = None
        self.storage_keys = frozenset()   # (storage pointer, device) of every parameter: what an optimizer must own to matter
        _NATIVES.add(self)

    def __deepcopy__(self, memo):
        return _Native()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.handle = None
        self.bound_key = None
        self.epoch = 0
        self.named_params = None
        self.storage_keys = frozenset()
        _NATIVES.add(self)

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().dsg_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


class UNet1D(nn.Module):

    def __init__(self, input_dim=3, proj_dim=16, cond_dim=4,
                 dims=(8, 4, 2),
                 is_attn=(False, False, False),
                 middle_attn=False,
                 n_blocks=2):
        super().__init__()
        if any(is_attn) or middle_attn:
            raise NotImplementedError("AttentionBlock is never enabled by the reference's call sites and is not built")
        self.cfg = dict(input_dim=int(input_dim), proj_dim=int(proj_dim), cond_dim=int(cond_dim),
                        dims=tuple(int(d) for d in dims), n_blocks=int(n_blocks))
        time_dim = proj_dim * 4
        n_res = len(dims)
        self.feature_proj = nn.Linear(input_dim, proj_dim)
        self.time_emb = TimeEmbedding(time_dim)

        down = []
        width = proj_dim
        for i in range(n_res):
            down += [DownBlock(width, width, time_dim, cond_dim) for _ in range(n_blocks)]
            down.append(Downsample(width, dims[i]))
            width = dims[i]
            if i == n_res - 1:
                down += [DownBlock(width, width, time_dim, cond_dim) for _ in range(n_blocks)]
        self.down = nn.ModuleList(down)
        self.middle = MiddleBlock(width, time_dim, cond_dim)
        up = []
        for i in reversed(range(n_res)):
            up += [UpBlock(width, width, time_dim, cond_dim) for _ in range(n_blocks + 1)]
            nxt = dims[i - 1] if i > 0 else proj_dim
            up.append(Upsample(width, nxt))
            width = nxt
            if i == 0:
                up += [UpBlock(width, width, time_dim, cond_dim) for _ in range(n_blocks + 1)]
        self.up = nn.ModuleList(up)
        self.norm = nn.LayerNorm(width)
        self.final = nn.Linear(width, input_dim)
        self._native = _Native()

    # ------------------------------------------------------------------ native plumbing
    def native_handle(self, twin=0):
        """Create (once) the dsg handle and (re)bind the current parameter tensors; returns the raw handle.
        twin > 0: a further handle of the SAME module (own workspace, own packed copy of the same weights): DDPM.forward runs the two
        halves of a very large training batch on two handles and two streams at once (ddpm.py, `train_split_min_rows`)."""
        L = _lib.lib()
        if twin:
            twins = self.__dict__.setdefault("_twins", {})
            nat = twins.get(twin)
            if nat is None:
                nat = twins[twin] = _Native()
                nat.epoch = self._native.epoch
        else:
            nat = self._native
        params = self._named_param_list()
        if not params[0][1].is_cuda:
            raise RuntimeError("UNet1D: parameters are not on a HIP device; libdiffsg_hip has no CPU path "
                               "(move the model with .to('cuda'))")
        if nat.handle is None:
            c = self.cfg
            d = _lib.UNetDesc()
            d.input_dim, d.proj_dim, d.cond_dim, d.n_blocks = c["input_dim"], c["proj_dim"], c["cond_dim"], c["n_blocks"]
            d.n_res = len(c["dims"])
            for i, v in enumerate(c["dims"]):
                d.dims[i] = v
            with torch.cuda.device(params[0][1].device):
                hd = L.dsg_create(ctypes.byref(d))
            if not hd:
                raise RuntimeError("libdiffsg_hip: " + L.dsg_last_error().decode())
            nat.handle = hd
            # every new handle starts with the module's settings: a twin, and also the primary handle of a deep-copied module, which
            # keeps `_settings` but gets a fresh _Native (its twin would otherwise run f32 beside a split_f16 primary)
            st = self.__dict__.get("_settings", {})
            if "precision" in st:
                _lib.check(L.dsg_set_precision(hd, st["precision"]))
            if "policy" in st:
                _lib.check(L.dsg_set_launch_policy(hd, *st["policy"]))
            for code, val in st.get("options", {}).items():
                _lib.check(L.dsg_set_option(hd, code, val))
            n = L.dsg_param_count(hd)
            names = [L.dsg_param_name(hd, i).decode() for i in range(n)]
            if names != [k for k, _ in params]:
                raise RuntimeError("state-dict layout mismatch between UNet1D and libdiffsg_hip")
            for i, (k, v) in enumerate(params):
                if v.numel() != L.dsg_param_numel(hd, i):
                    raise RuntimeError(f"size mismatch for {k}")
        if twin:
            nat.storage_keys = self._native.storage_keys        # so that an optimizer step reaches the twin's epoch too
        key = (nat.epoch,) + tuple((v.data_ptr(), v._version) for _, v in params)
        if key != nat.bound_key:
            for k, v in params:
                if v.dtype != torch.float32 or not v.is_contiguous():
                    raise RuntimeError(f"{k}: parameters must be contiguous float32")
            arr = (ctypes.c_void_p * len(params))(*[v.data_ptr() for _, v in params])
            with torch.cuda.device(params[0][1].device):
                _lib.check(L.dsg_bind_weights(nat.handle, arr, len(params), _lib.stream_ptr()))
            nat.bound_key = key
        return nat.handle

    def _named_param_list(self):
        """[(state-dict key, Parameter)] in state-dict order, cached: walking the module tree costs more host time per
        training step than the whole launch sequence.  The Parameter OBJECTS are stable under .to(), load_state_dict and
        train.flatten_parameters (they swap `.data`); `_apply` drops the cache anyway, and code that assigns new Parameter
        objects into sub-modules must call `mark_weights_changed()`."""
        nat = self._native
        if getattr(nat, 'named_params', None) is None:
            nat.named_params = list(self.state_dict(keep_vars=True).items())
            nat.storage_keys = frozenset((v.untyped_storage().data_ptr(), v.device) for _, v in nat.named_params)
            _OPT_OWNERS.clear()            # ownership is decided per (optimizer, parameter storage): recompute
        return nat.named_params

    def param_list(self):
        """The parameter tensors in state-dict order (cached, see `_named_param_list`)."""
        return [v for _, v in self._named_param_list()]

    def _apply(self, fn, *a, **k):
        self._native.named_params = None
        return super()._apply(fn, *a, **k)

    def mark_weights_changed(self, rebuild=False):
        """Parameters were updated in place through an aliasing tensor (train.FlatAdam): re-pack on the next call.
        rebuild=True also drops the cached parameter list (new Parameter objects were assigned into sub-modules)."""
        self._native.epoch += 1
        for nat in self.__dict__.get("_twins", {}).values():
            nat.epoch += 1
        if rebuild:
            self._native.named_params = None

    def set_precision(self, mode):
        """Arithmetic of the >= 64-wide blocks inside DDPM.sample: "split_f16" (default; float32-accurate hi/lo fp16
        split on the f16 matrix cores, f32 accumulate) or "f32" (exact v_mfma_f32_32x32x2_f32)."""
        code = {"split_f16": 0, "f32": 1}[mode]
        for hd in self._all_handles():
            _lib.check(_lib.lib().dsg_set_precision(hd, code))
        self.__dict__.setdefault("_settings", {})["precision"] = code

    @property
    def precision(self):
        """The handle's current arithmetic mode ("split_f16" unless `set_precision` changed it)."""
        return {0: "split_f16", 1: "f32"}[self.__dict__.get("_settings", {}).get("precision", 0)]

    def range_exceeded(self):
        """True if, since the last query, a raw operand of the split-f16 path left fp16's range (dsg_range_status_stream:
        synchronises torch's CURRENT stream of the model's device -- the stream the launches went to -- and clears the flag; other
        streams keep running).  Outputs computed meanwhile are then wrong."""
        hit = False
        # every handle that EXISTS (a module that has not launched anything has no flag to read; the twin handle of a split training
        # step has a flag word of its own)
        for nat in [self._native] + list(self.__dict__.get("_twins", {}).values()):
            hd = nat.handle
            if hd is None:
                continue
            flag = ctypes.c_int(0)
            with torch.cuda.device(self._named_param_list()[0][1].device):
                _lib.check(_lib.lib().dsg_range_status_stream(hd, ctypes.byref(flag), _lib.stream_ptr()))
            hit = hit or bool(flag.value)
        return hit

    def check_range(self):
        """Raise if `range_exceeded()`: the caller should switch to `set_precision("f32")` and repeat the call."""
        if self.range_exceeded():
            raise RuntimeError("libdiffsg_hip: an activation exceeded the fp16 range of the split-f16 path (|x| > 6e4); "
                               "the results of the last calls are invalid -- use set_precision('f32') for this model")

    def set_launch_policy(self, coop_max_tiles=-1, narrow_small_max_tiles=-1):
        """Kernel forms by launch size (dsg_set_launch_policy): launches of at most `coop_max_tiles` 32-row tiles run
        the wide blocks cooperatively, at most `narrow_small_max_tiles` the small-launch narrow run; -1 = default,
        0 = always the large-launch forms (what a 65 536-row call uses)."""
        for hd in self._all_handles():
            _lib.check(_lib.lib().dsg_set_launch_policy(hd, int(coop_max_tiles), int(narrow_small_max_tiles)))
        self.__dict__.setdefault("_settings", {})["policy"] = (int(coop_max_tiles), int(narrow_small_max_tiles))

    def set_option(self, name, value):
        """Per-handle kernel-form switches (dsg_set_option): "narrow_valu8" -- the 8-wide bottom of the net on the vector
        unit in float32 inside large sampling launches (default on); "train_time_beside" -- the time-path backward of large
        training steps on the side stream beside the last weight-gradient launch (default on); "wgrad_narrow_part" -- the narrow
        run's weight gradients as a third early part (default off); "tile_step" -- small launches run a denoiser pass as feature_proj + one
        launch that walks every operator per row tile (default on; csrc/dsg_tile.hpp); "panel_half" -- half-size weight panels in the
        persistent 128-wide kernels (default on); "f32_pair" -- exact path: a wide block and its consuming Linear in one launch (default on)."""
        code = {"narrow_valu8": 1, "train_time_beside": 2, "wgrad_narrow_part": 4, "tile_step": 8, "panel_half": 16, "f32_pair": 32}[name]
        for hd in self._all_handles():
            _lib.check(_lib.lib().dsg_set_option(hd, code, int(value)))
        self.__dict__.setdefault("_settings", {}).setdefault("options", {})[code] = int(value)

    def _all_handles(self):
        """The primary handle and every twin that exists (per-handle settings apply to all of them)."""
        return [self.native_handle()] + [self.native_handle(k) for k in self.__dict__.get("_twins", {})]

    def forward(self, x, t, cond, cond_mask):
        """
        :param x: (batch_size, input_dim)
        :param t: (1, batch_size) time value already divided by T
        :param cond: (batch_size, cond_dim)
        :param cond_mask: (batch_size, 1)
        :return: estimated noise (batch_size, input_dim).  Inference only: gradients flow through DDPM.forward.
        """
        hd = self.native_handle()
        B = x.shape[0]
        x = x.detach().to(torch.float32).contiguous()
        cond = cond.detach().to(torch.float32).contiguous()
        tv = t.detach().to(torch.float32).reshape(-1).contiguous()
        mk = cond_mask.detach().to(torch.float32).reshape(-1).contiguous()
        if tv.numel() != B or mk.numel() != B or cond.shape[0] != B:
            raise ValueError("t, cond and cond_mask must have one entry per row of x")
        if x.shape[1] != self.cfg["input_dim"] or cond.shape[1] != self.cfg["cond_dim"]:
            raise ValueError("x / cond feature size does not match the model")
        out = torch.empty_like(x)
        if B == 0:
            return out                     # no rows, nothing to launch (the reference returns an empty tensor too)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().dsg_unet_forward(hd, _lib.ptr(x), _lib.ptr(tv), _lib.ptr(cond), _lib.ptr(mk),
                                                   _lib.ptr(out), B, _lib.stream_ptr()))
        return out
