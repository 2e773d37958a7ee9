"""Minimal host-side frame container (field layout of the reference's Frame / Keyframes,
/root/reference/src/mlx_mast3r_slam/frame.py:27-143, :146-262).

Out of the hot path (SURVEY §8 marks frame.py as host state): only the fields the
operator API touches are kept, as torch tensors on the ROCm device.  Pointmap fusion
supports the reference default "weighted_pointmap" (:118-123) plus "first"/"recent";
the fused on-device version is a §8f "next" item.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


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
    filtering_mode: str = "weighted_pointmap"

    def __post_init__(self):
        if self.T_WC is None:
            self.T_WC = identity_pose(self.img.device if self.img.is_cuda else "cuda")

    def update_pointmap(self, X: torch.Tensor, C: torch.Tensor) -> None:
        """frame.py:75-133 (modes first / recent / weighted_pointmap)."""
        X = X.reshape(-1, 3)
        C = C.reshape(-1, 1)
        if self.N == 0:
            self.X_canon, self.C, self.N, self.N_updates = X, C, 1, 1
            return
        mode = self.filtering_mode
        if mode == "first":
            if self.N_updates == 1:
                self.X_canon, self.C, self.N = X, C, 1
        elif mode == "recent":
            self.X_canon, self.C, self.N = X, C, 1
        elif mode == "weighted_pointmap":
            total = self.C + C
            self.X_canon = (self.C * self.X_canon + C * X) / total
            self.C = total
            self.N += 1
        else:
            raise NotImplementedError(f"filtering_mode {mode!r} is outside the hot-path scope")
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


def create_frame(frame_id: int, img: torch.Tensor, T_WC: Optional[torch.Tensor] = None) -> Frame:
    """frame.py:299-343 (subset): img [3,H,W] float [0,1] or uint8 [H,W,3]."""
    if img.dim() != 3:
        raise ValueError("img must be [3,H,W] or [H,W,3]")
    h, w = (img.shape[1], img.shape[2]) if img.shape[0] == 3 else (img.shape[0], img.shape[1])
    shape = torch.tensor([[h, w]], dtype=torch.int32)
    return Frame(frame_id=frame_id, img=img, img_shape=shape, img_true_shape=shape.clone(), T_WC=T_WC)
