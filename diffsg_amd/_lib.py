"""ctypes binding of libdiffsg_hip.so (C ABI: include/diffsg.h).

There is deliberately no fallback: if the HIP library cannot be loaded, or no MI355X is visible, every compute entry
point raises.  PyTorch is used only for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffsg_hip.so")
RES_PATH = os.path.join(_HERE, "libdiffsg_hip.resources.txt")     # hipcc's kernel-resource-usage remarks of the build (build())
SOURCES = [os.path.join(_HERE, "csrc", "dsg_api.hip")]


def _headers():
    """Everything dsg_api.hip includes: every header under csrc/ and the C-ABI declaration."""
    import glob
    return (sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hpp")) + glob.glob(os.path.join(_HERE, "csrc", "*.inc")))
            + [os.path.join(os.path.dirname(_HERE), "include", "diffsg.h")])

_ID_MARK = b"DSG_BUILD_ID="


def source_id(extra_flags: str = "") -> str:
    """sha256 over the sources the library is built from (file names + contents) and any extra compiler flags: compiled into
    the library as its build id.  A measurement binary (DSG_EXTRA_CXXFLAGS=-DDSG_CYCLE_STAMPS ...) therefore never carries the
    id of the production build of the same tree: `_stale()` compares with the flag-less id and rebuilds over it."""
    import hashlib
    h = hashlib.sha256()
    for p in SOURCES + _headers():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    if extra_flags.split():
        h.update(b"flags\0" + " ".join(extra_flags.split()).encode())
    return h.hexdigest()


def built_id():
    """The build id compiled into libdiffsg_hip.so (read from the file, nothing is loaded), or None."""
    if not os.path.exists(LIB_PATH):
        return None
    with open(LIB_PATH, "rb") as f:
        blob = f.read()
    i = blob.find(_ID_MARK)
    return blob[i + len(_ID_MARK): i + len(_ID_MARK) + 64].decode("ascii", "replace") if i >= 0 else None

MAX_RES = 8


class UNetDesc(ctypes.Structure):
    _fields_ = [("input_dim", ctypes.c_int), ("proj_dim", ctypes.c_int), ("cond_dim", ctypes.c_int),
                ("n_res", ctypes.c_int), ("dims", ctypes.c_int * MAX_RES), ("n_blocks", ctypes.c_int)]


def _stale() -> bool:
    """The library is missing, or was built from other sources than the ones in the tree (content hash, not mtimes: the
    binary travels to the GPU box as a file copy)."""
    return built_id() != source_id(os.environ.get("DSG_EXTRA_CXXFLAGS", ""))   # a process without the flags refuses a measurement build


def _resources_current() -> bool:
    try:
        with open(RES_PATH) as f:
            return f.readline().strip() == "build_id " + source_id(os.environ.get("DSG_EXTRA_CXXFLAGS", ""))
    except OSError:
        return False


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (cross-compiles without a GPU).  Serialised by a file lock: the ranks of a
    multi-process launch must not compile the same file at once."""
    if not force and not _stale() and _resources_current():
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not force and not _stale() and _resources_current():      # another process built it while this one waited
                return LIB_PATH
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            # -fno-slp-vectorize: with the SLP vectoriser on, the packed-f32 code it forms in k_wgrad_h gave run-to-run
            # different results on gfx950 ("Build flags"); without it the library is deterministic and 2-5 % faster.
            tmp = LIB_PATH + f".{os.getpid()}.tmp"
            extra = os.environ.get("DSG_EXTRA_CXXFLAGS", "")
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-fno-slp-vectorize",
                   # the panel kernel's 48-slot MFMA loops must unroll completely (the slots index register arrays): the default
                   # limit on `#pragma unroll` (16 K) is below the largest variant, whose arrays then land in scratch memory
                   "-mllvm", "-pragma-unroll-threshold=200000",
                   # per-kernel register / scratch / LDS figures of THIS build (remarks on stderr), kept beside the library:
                   # tests/test_host_api.py fails on any scratch in a kernel of the 65 536-row step (kernel_resources())
                   "-Rpass-analysis=kernel-resource-usage",
                   f'-DDSG_BUILD_ID_STR="{source_id(extra)}"', "-o", tmp] + extra.split() + SOURCES
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            remarks = [ln for ln in r.stderr.splitlines() if "remark:" in ln]
            if r.returncode != 0:
                print("\n".join(ln for ln in r.stderr.splitlines() if "remark:" not in ln), file=sys.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            warn = [ln for ln in r.stderr.splitlines() if "warning:" in ln]      # (a remark drags its source line and caret along: dropped)
            if warn:
                print("\n".join(warn), file=sys.stderr)
            with open(RES_PATH + ".tmp", "w") as f:
                f.write("build_id " + source_id(extra) + "\n" + "\n".join(remarks) + "\n")
            os.replace(RES_PATH + ".tmp", RES_PATH)
            os.replace(tmp, LIB_PATH)           # atomic: a reader never maps a half-written library
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB_PATH


def kernel_resources():
    """{demangled kernel name: {"vgprs", "agprs", "scratch", "occupancy", "lds", "sgprs"}} of the library in the tree, from the
    remarks hipcc printed when `build()` compiled it (-Rpass-analysis=kernel-resource-usage).  Raises if the record does not
    belong to the current sources (run build() first)."""
    import re
    if not os.path.exists(RES_PATH):
        raise RuntimeError(f"{RES_PATH} is missing: run build()")
    with open(RES_PATH) as f:
        txt = f.read()
    first, _, rest = txt.partition("\n")
    if first.strip() != "build_id " + source_id(os.environ.get("DSG_EXTRA_CXXFLAGS", "")):
        raise RuntimeError(f"{RES_PATH} was written by a build of other sources: run build()")
    blocks = re.split(r"remark: [^\n]*Function Name: ", rest)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    except Exception:
        dem = names
    out = {}
    for b, n in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*", "", n).strip()
        out[n] = {"vgprs": g("VGPRs"), "agprs": g("AGPRs"), "scratch": g(r"ScratchSize \[bytes/lane\]"),
                  "occupancy": g(r"Occupancy \[waves/SIMD\]"), "lds": g(r"LDS Size \[bytes/block\]"), "sgprs": g("SGPRs")}
    return out


_lib = None

_SIGS = {
    # name: (restype, argtypes)
    "dsg_create": (ctypes.c_void_p, [ctypes.POINTER(UNetDesc)]),
    "dsg_destroy": (None, [ctypes.c_void_p]),
    "dsg_last_error": (ctypes.c_char_p, []),
    "dsg_param_count": (ctypes.c_int, [ctypes.c_void_p]),
    "dsg_param_name": (ctypes.c_char_p, [ctypes.c_void_p, ctypes.c_int]),
    "dsg_param_numel": (ctypes.c_longlong, [ctypes.c_void_p, ctypes.c_int]),
    "dsg_param_total": (ctypes.c_longlong, [ctypes.c_void_p]),
    "dsg_train_step": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int, ctypes.c_void_p]),
    "dsg_train_draws": (ctypes.c_int, [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dsg_train_step_seeded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_ulonglong,
                                             ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_void_p]),
    "dsg_train_step_seeded_dyn": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p,
                                                 ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_int, ctypes.c_void_p]),
    "dsg_adam_step_dyn": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p,
                                         ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "dsg_bind_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]),
    "dsg_set_precision": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "dsg_set_launch_policy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "dsg_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "dsg_set_renorm_hook": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dsg_range_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "dsg_range_status_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "dsg_build_id": (ctypes.c_char_p, []),
    "dsg_reserve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "dsg_train_profile_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "dsg_train_profile": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "dsg_unet_forward": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]),
    "dsg_row_softmax": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    "dsg_msr_decode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    "dsg_co_decode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    "dsg_msr_rate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    "dsg_co_cost": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    "dsg_nu_decode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_float, ctypes.c_void_p]),
    "dsg_nu_rate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    "dsg_sum_rate_gen": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_void_p]),
    "dsg_co_minlp_search": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                                           ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]),
    "dsg_sample": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong,
                                  ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p]),
    "dsg_sample_rec": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong,
                                      ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dsg_sample_chunked": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong),
                                          ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p]),
    "dsg_adam_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_longlong,
                                     ctypes.c_void_p]),
    "dsg_ema_update": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_longlong,
                                      ctypes.c_void_p]),
    "dsg_fused_range": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "dsg_op_count": (ctypes.c_int, [ctypes.c_void_p]),
    "dsg_op_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                   ctypes.POINTER(ctypes.c_double)]),
    "dsg_op_profile": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_int)]),
    "dsg_box_calibrate": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]),
    "dsg_time_op": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                   ctypes.c_void_p]),
}


RENORM_REDUCE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


def exported_symbols():
    return sorted(_SIGS)


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback for this path")
        if _stale():
            # never compile here: this process may already have touched the GPU, and a silent rebuild would hide the mismatch
            raise RuntimeError(f"{LIB_PATH} was built from other sources than the ones in the tree (build id {built_id()} != "
                               f"{source_id()}): run `python -c 'import __graft_entry__ as g; g.build()'` first")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export what diffsg.h declares
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise RuntimeError("libdiffsg_hip: " + lib().dsg_last_error().decode())


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
