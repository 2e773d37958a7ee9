"""Dense matching front end on the MI355X.

Mirror of /root/reference/src/mlx_mast3r_slam/matching.py (match :12, match_simple :41,
pixel_to_lin :93, lin_to_pixel :106, prep_for_iter_proj :134, match_iterative_proj :339)
over torch tensors on the ROCm device.  Every function is batched over B pairs and
stream-ordered; nothing synchronises with the host.
"""
from __future__ import annotations

import torch

from . import _ffi
from .config import get_config
from . import kernels


def pixel_to_lin(p: torch.Tensor, w: int) -> torch.Tensor:
    """matching.py:93-103."""
    return p[..., 0] + w * p[..., 1]


def lin_to_pixel(idx: torch.Tensor, w: int) -> torch.Tensor:
    """matching.py:106-118."""
    return torch.stack([idx % w, idx // w], dim=-1)


def _check_maps(X11, X21):
    X11 = _ffi.check(X11, torch.float32, "X11")
    if X11.dim() != 4 or X11.shape[-1] != 3:
        raise ValueError(f"X11 must be [B,H,W,3], got {tuple(X11.shape)}")
    b, h, w, _ = X11.shape
    X21 = _ffi.check(X21, torch.float32, "X21")
    if X21.numel() != b * h * w * 3:
        raise ValueError(f"X21 must hold {b}x{h}x{w}x3 values, got {tuple(X21.shape)}")
    return X11, X21, b, h, w


def _check_idx(idx, b, n):
    if idx is None:
        return None
    idx = _ffi.check(idx, torch.int64, "idx_1_to_2_init") if idx.dtype == torch.int64 else \
        _ffi.check(idx.to(torch.int64), torch.int64, "idx_1_to_2_init")
    if tuple(idx.shape) != (b, n):
        raise ValueError(f"idx_1_to_2_init must be [{b},{n}], got {tuple(idx.shape)}")
    return idx


def prep_for_iter_proj(X11, X21, idx_1_to_2_init=None):
    """matching.py:134-175 -> (rays_with_grad [B,H,W,9], pts3d_norm [B,N,3], p_init [B,N,2])."""
    X11, X21, b, h, w = _check_maps(X11, X21)
    n = h * w
    idx = _check_idx(idx_1_to_2_init, b, n)
    dev = X11.device
    rays = torch.empty((b, h, w, 9), dtype=torch.float32, device=dev)
    tgt = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
    p0 = torch.empty((b, n, 2), dtype=torch.float32, device=dev)
    _ffi.call("m3_prep_iter_proj", _ffi.ptr(X11), _ffi.ptr(X21), _ffi.ptr(idx), _ffi.ptr(rays), _ffi.ptr(tgt),
              _ffi.ptr(p0), b, h, w, _ffi.stream_ptr())
    return rays, tgt, p0


def match_simple(X11, X21, D11=None, D21=None, idx_1_to_2_init=None):
    """matching.py:41-90 -> (idx [B,N] int64, valid [B,N,1] bool)."""
    cfg = get_config()["matching"]
    X11, X21, b, h, w = _check_maps(X11, X21)
    n = h * w
    idx = _check_idx(idx_1_to_2_init, b, n)
    out = torch.empty((b, n), dtype=torch.int64, device=X11.device)
    valid = torch.empty((b, n), dtype=torch.uint8, device=X11.device)
    _ffi.call("m3_match_simple", _ffi.ptr(X11), _ffi.ptr(X21), _ffi.ptr(idx), _ffi.ptr(out), _ffi.ptr(valid),
              b, h, w, float(cfg["dist_thresh"]), _ffi.stream_ptr())
    return out, valid.bool()[:, :, None]


def match_iterative_proj(X11, X21, D11, D21, idx_1_to_2_init=None, *, stop_scope: str = "batch"):
    """matching.py:339-461: prep -> iter_proj -> refine_matches -> occlusion test -> linear index.

    Returns (idx_1_to_2 [B,N] int64, valid [B,N,1] bool).  stop_scope="batch" makes a batched
    call equal to B separate calls (the reference only ever calls this with B=1)."""
    cfg = get_config()["matching"]
    X11, X21, b, h, w = _check_maps(X11, X21)
    n = h * w
    rays, tgt, p0 = prep_for_iter_proj(X11, X21, idx_1_to_2_init)
    p1, valid_proj = kernels.iter_proj(rays, tgt, p0, cfg["max_iter"], cfg["lambda_init"],
                                       cfg["convergence_thresh"], stop_scope=stop_scope)
    dev = X11.device
    vp = valid_proj.to(torch.uint8)
    p_int = None
    if cfg.get("use_refine", True) and cfg.get("refine_radius", 3) > 0:
        D11 = _ffi.check(D11, torch.float32, "D11")
        D21 = _ffi.check(D21, torch.float32, "D21")
        d = D11.shape[-1]
        p_trunc = torch.empty((b, n, 2), dtype=torch.int32, device=dev)
        _ffi.call("m3_trunc_i32", _ffi.ptr(p1), _ffi.ptr(p_trunc), b * n * 2, _ffi.stream_ptr())
        p_int = kernels.refine_matches(D11.reshape(b, h, w, d), D21.reshape(b, n, d), p_trunc,
                                       cfg.get("refine_radius", 3), cfg.get("refine_dilation", 2),
                                       chained=cfg.get("refine_chained", False))
    idx = torch.empty((b, n), dtype=torch.int64, device=dev)
    valid = torch.empty((b, n), dtype=torch.uint8, device=dev)
    _ffi.call("m3_match_epilogue", _ffi.ptr(X11), _ffi.ptr(X21), _ffi.ptr(p_int), _ffi.ptr(p1), _ffi.ptr(vp),
              _ffi.ptr(idx), _ffi.ptr(valid), b, h, w, float(cfg["dist_thresh"]), _ffi.stream_ptr())
    return idx, valid.bool()[:, :, None]


def match(X11, X21, D11, D21, idx_1_to_2_init=None):
    """matching.py:12-38: dispatch on config matching.use_simple (default True)."""
    if get_config().get("matching", {}).get("use_simple", True):
        return match_simple(X11, X21, D11, D21, idx_1_to_2_init)
    return match_iterative_proj(X11, X21, D11, D21, idx_1_to_2_init)
