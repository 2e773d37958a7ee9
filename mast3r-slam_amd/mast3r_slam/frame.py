"""Frame / keyframe state (field layout of the reference's Frame / Keyframes,
/root/reference/src/mlx_mast3r_slam/frame.py:27-143, :146-262), tensors resident on the ROCm device.

Pointmap fusion (all six filtering modes of frame.py:75-133) runs in ONE HIP kernel, in place
on buffers the frame owns (m3_fuse_pointmap, csrc/frame.hip), optionally with the Sim3.act of the
tracker's keyframe update fused in.  SURVEY section 8f rank 3.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import _ffi
from .config import get_config

FUSE_REPLACE, FUSE_INDEP_CONF, FUSE_WEIGHTED_POINTMAP, FUSE_WEIGHTED_SPHERICAL = 0, 1, 2, 3   # include/m3slam.h
_MODES = ("first", "recent", "best_score", "indep_conf", "weighted_pointmap", "weighted_spherical")


def identity_pose(device="cuda") -> torch.Tensor:
    """Sim3.identity().data, liegroups/sim3.py:37-55 -> [1,8]."""
    return torch.tensor([[0, 0, 0, 0, 0, 0, 1, 1]], dtype=torch.float32, device=device)


@dataclass
class Frame:
    frame_id: int
    img: torch.Tensor                       # [3,H,W] float in [0,1] (or uint8 [H,W,3])
    img_shape: Optional[torch.Tensor] = None
    img_true_shape: Optional[torch.Tensor] = None
    T_WC: Optional[torch.Tensor] = None     # [1,8]
    X_canon: Optional[torch.Tensor] = None  # [N,3]
    C: Optional[torch.Tensor] = None        # [N,1]
    feat: Optional[torch.Tensor] = None
    pos: Optional[torch.Tensor] = None
    N: int = 0
    N_updates: int = 0
    K: Optional[torch.Tensor] = None
    filtering_mode: Optional[str] = None    # None: config tracking.filtering_mode (frame.py:82-83)
    _best: Optional[torch.Tensor] = None     # "best_score" mode: device float [2] = (best score, last update replaced)

    @property
    def _score(self) -> Optional[float]:
        """frame.py's host-side `_score` (read-only view of the device state; one synchronisation per read)."""
        return None if self._best is None else float(self._best[0])

    @_score.setter
    def _score(self, value: Optional[float]) -> None:
        if value is None:
            self._best = None
        else:
            dev = self._best.device if self._best is not None else (self.img.device if self.img.is_cuda else "cuda")
            self._best = torch.tensor([float(value), 0.0], dtype=torch.float32, device=dev)

    def __post_init__(self):
        if self.T_WC is None:
            self.T_WC = identity_pose(self.img.device if self.img.is_cuda else "cuda")

    def score_tensor(self, C: torch.Tensor) -> torch.Tensor:
        """frame.py:59-73 on the device: float32 [1] = the median (mean of the two middle values for an even count, as
        mx.median / np.median) or the mean of C - m3_median_f32 is an exact radix select, no sort and no host sync."""
        v = _ffi.check(C.reshape(-1), torch.float32, "C")
        if get_config()["tracking"].get("filtering_score", "median") != "median":
            return v.mean().reshape(1)
        out = torch.empty(1, dtype=torch.float32, device=v.device)
        ws = torch.empty(int(_ffi.lib().m3_median_ws_words()), dtype=torch.int32, device=v.device)
        _ffi.call("m3_median_f32", _ffi.ptr(v), v.numel(), _ffi.ptr(ws), _ffi.ptr(out), _ffi.stream_ptr())
        return out

    def get_score(self, C: torch.Tensor) -> float:
        """frame.py:59-73 as the reference returns it: a host float (ONE synchronisation; update_pointmap itself does not
        call this - its "best_score" decision stays on the device)."""
        return float(self.score_tensor(C))

    def update_pointmap(self, X: torch.Tensor, C: torch.Tensor, T: Optional[torch.Tensor] = None) -> None:
        """frame.py:75-133.  X [N,3] (any shape with N*3 elements), C [N,1]; `T` ([1,8] Sim3, optional)
        moves X first (tracker.py:146-147).  The mode comes from config tracking.filtering_mode unless
        the frame overrides it."""
        X = _ffi.check(X.reshape(-1, 3), torch.float32, "X")
        C = _ffi.check(C.reshape(-1, 1), torch.float32, "C", (X.shape[0], 1))
        if T is not None:
            T = _ffi.check(T.reshape(-1), torch.float32, "T", (8,))
        mode = self.filtering_mode or get_config()["tracking"].get("filtering_mode", "weighted_pointmap")
        if mode not in _MODES:
            raise ValueError(f"unknown filtering_mode {mode!r}")
        n = X.shape[0]

        def fuse(kind):
            _ffi.call("m3_fuse_pointmap", _ffi.ptr(self.X_canon), _ffi.ptr(self.C), _ffi.ptr(X), _ffi.ptr(C),
                      _ffi.ptr(T), n, kind, _ffi.stream_ptr())

        if self.N == 0:
            self.X_canon = torch.empty((n, 3), dtype=torch.float32, device=X.device)    # buffers the frame owns
            self.C = torch.empty((n, 1), dtype=torch.float32, device=X.device)
            fuse(FUSE_REPLACE)
            self.N, self.N_updates = 1, 1
            if mode == "best_score":                      # device state: (best score so far, replaced-flag of the last update)
                self._best = torch.cat([self.score_tensor(C), torch.ones(1, dtype=torch.float32, device=X.device)])
            return
        if self.X_canon.shape[0] != n:
            raise ValueError("pointmap size changed")
        if mode == "first":
            if self.N_updates == 1:
                fuse(FUSE_REPLACE); self.N = 1
        elif mode == "recent":
            fuse(FUSE_REPLACE); self.N = 1
        elif mode == "best_score":                        # frame.py:103-107: winner takes all, decided on the device
            if self._best is None:
                self._best = torch.zeros(2, dtype=torch.float32, device=X.device)
            score = self.score_tensor(C)
            _ffi.call("m3_fuse_pointmap_if_better", _ffi.ptr(self.X_canon), _ffi.ptr(self.C), _ffi.ptr(X), _ffi.ptr(C),
                      _ffi.ptr(T), n, _ffi.ptr(score), _ffi.ptr(self._best), _ffi.stream_ptr())
            # frame.py:103-109 resets N only when the new score WINS.  In a pure best_score run N is 1 already (no
            # sync); after weighted_* updates (a per-frame mode override) N > 1 and the device's verdict is read once.
            if self.N != 1 and bool(self._best[1] > 0):
                self.N = 1
        elif mode == "indep_conf":
            fuse(FUSE_INDEP_CONF); self.N = 1
        elif mode == "weighted_pointmap":
            fuse(FUSE_WEIGHTED_POINTMAP); self.N += 1
        else:
            fuse(FUSE_WEIGHTED_SPHERICAL); self.N += 1
        self.N_updates += 1

    def get_average_conf(self) -> Optional[torch.Tensor]:
        """frame.py:135-143."""
        return None if self.C is None else self.C / self.N


class Keyframes:
    """List-backed keyframe store (frame.py:146-262, subset)."""

    def __init__(self) -> None:
        self._frames: list[Frame] = []

    def __len__(self) -> int:
        return len(self._frames)

    def __getitem__(self, i: int) -> Frame:
        return self._frames[i]

    def __setitem__(self, i: int, f: Frame) -> None:
        self._frames[i] = f

    def append(self, f: Frame) -> None:
        self._frames.append(f)

    def last_keyframe(self) -> Optional[Frame]:
        return self._frames[-1] if self._frames else None

    def update_T_WCs(self, T_WCs: torch.Tensor, idx) -> None:
        """frame.py:236-246: write optimised poses ([M,8]) back to keyframes idx ([M])."""
        ids = idx.tolist() if isinstance(idx, torch.Tensor) else list(idx)
        for row, k in zip(T_WCs.reshape(-1, 8), ids):
            self._frames[int(k)].T_WC = row.reshape(1, 8).clone()

    def set_intrinsics(self, K: torch.Tensor) -> None:
        self.K = K
        for f in self._frames:
            f.K = K

    def get_intrinsics(self) -> Optional[torch.Tensor]:
        return getattr(self, "K", None)


def create_frame(frame_id: int, img: torch.Tensor, T_WC: Optional[torch.Tensor] = None) -> Frame:
    """frame.py:299-343 (subset): img [3,H,W] float [0,1] or uint8 [H,W,3]."""
    if img.dim() != 3:
        raise ValueError("img must be [3,H,W] or [H,W,3]")
    h, w = (img.shape[1], img.shape[2]) if img.shape[0] == 3 else (img.shape[0], img.shape[1])
    shape = torch.tensor([[h, w]], dtype=torch.int32)
    return Frame(frame_id=frame_id, img=img, img_shape=shape, img_true_shape=shape.clone(), T_WC=T_WC)
