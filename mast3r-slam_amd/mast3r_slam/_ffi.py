"""ctypes binding of libm3slam_hip.so (the C ABI declared in include/m3slam.h).

The library is the product: there is NO fallback.  If it is missing or a call
returns a non-zero status a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# M3SLAM_LIB: another build of the same library (experiments: ablation builds of one kernel file linked with the shipped objects)
LIB_PATH = os.environ.get("M3SLAM_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libm3slam_hip.so")
HEADERS = [os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", h)
           for h in ("m3slam.h", "m3slam_model.h")]

_lib = None

_CT = {
    "int": C.c_int, "float": C.c_float, "int64_t": C.c_int64, "double": C.c_double,
    "void": None,
}


def declared_symbols() -> list[str]:
    """Every function name declared in include/*.h (used by the CPU export test)."""
    names = []
    for h in HEADERS:
        if not os.path.exists(h):
            continue
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(m3_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def _parse_prototypes():
    protos = {}
    for h in HEADERS:
        if not os.path.exists(h):
            continue
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        for m in re.finditer(r"(const char \*|int64_t|int|void)\s*(m3_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
            ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
            argt = []
            if args and args != "void":
                for a in args.split(","):
                    a = a.strip()
                    if "*" in a:
                        argt.append(C.c_void_p)
                    else:
                        base = a.replace("const ", "").split()[0]
                        argt.append(_CT[base])
            rt = C.c_char_p if ret.startswith("const char") else _CT[ret]
            protos[name] = (rt, argt)
    return protos


def lib():
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python mast3r-slam_amd/build.py` "
                "(the HIP library is the product path; there is no fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (rt, argt) in _parse_prototypes().items():
            fn = getattr(L, name)
            fn.restype = rt
            fn.argtypes = argt
        _lib = L
    return _lib


class GemmDesc(C.Structure):
    """m3_gemm_desc of include/m3slam_model.h (field order and types must match the header)."""
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("W1", C.c_void_p), ("bias", C.c_void_p), ("bias1", C.c_void_p),
                ("C", C.c_void_p), ("R", C.c_void_p), ("rope_pos", C.c_void_p), ("c16", C.c_void_p), ("stats_out", C.c_void_p),
                ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_colsum1", C.c_void_p), ("r_lo", C.c_void_p), ("c_lo", C.c_void_p),
                ("a_gstride", C.c_int64), ("c_gstride", C.c_int64), ("ln_gstride", C.c_int64), ("stats_gstride", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ldc", C.c_int32), ("epilogue", C.c_int32),
                ("dtype", C.c_int32), ("groups", C.c_int32), ("tokens_per_image", C.c_int32), ("rope_cols", C.c_int32),
                ("q_cols", C.c_int32), ("ln_slots", C.c_int32),
                ("rope_base", C.c_float), ("q_scale", C.c_float), ("ln_eps", C.c_float), ("stats_slots", C.c_int32), ("rope_max_pos", C.c_int32)]


class PackSeg(C.Structure):
    """m3_pack_seg of include/m3slam_model.h."""
    _fields_ = [("src", C.c_void_p), ("dst_off", C.c_int64), ("nbytes", C.c_int64), ("mode", C.c_int32), ("reserved", C.c_int32)]


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def check(t: torch.Tensor, dtype, name: str, shape=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: must live on the ROCm device (got {t.device}); no CPU path exists")
    if t.dtype != dtype and not (isinstance(dtype, tuple) and t.dtype in dtype):
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
    return t.contiguous()


# When set to a dict, every `call` of an entry point listed in PROFILE_NAMES appends a (start, end) pair of
# events recorded on the current stream around the launch sequence of that call (bench.py's per-kernel-family
# device times of the matcher and the Gauss-Newton solve).
PROFILE = None
PROFILE_NAMES = ()


def call(name: str, *args):
    """Call an int-returning entry point; non-zero status -> RuntimeError."""
    L = lib()
    prof = PROFILE is not None and name in PROFILE_NAMES
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = getattr(L, name)(*args)
    if prof:
        e1.record()
        PROFILE.setdefault(name, []).append((e0, e1))
    if rc != 0:
        msg = L.m3_status_string(rc).decode()
        herr = L.m3_last_hip_error().decode()
        raise RuntimeError(f"{name} failed with status {rc}: {msg}" + (f" [{herr}]" if herr else ""))
