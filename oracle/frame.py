"""CPU oracle (numpy) of the frame / keyframe state updates.  TEST INFRASTRUCTURE ONLY.

Restates, as text, /root/reference/src/mlx_mast3r_slam:
  * Frame.update_pointmap        frame.py:75-133   (modes first / recent / best_score / indep_conf /
                                                     weighted_pointmap / weighted_spherical)
  * Frame.get_average_conf       frame.py:135-143
  * cartesian_to_spherical / spherical_to_cartesian   geometry.py:318-351
  * the new-keyframe statistics  tracker.py:149-160 (match_frac_k, unique_frac_f)
The reference module imports mlx (absent here), so this part of the oracle is pinned by analytic
known-answer tests only ("parity unpinned" against reference outputs).
"""
from __future__ import annotations

import numpy as np


def cartesian_to_spherical(P):
    r = np.sqrt(np.sum(P * P, axis=-1, keepdims=True) + P.dtype.type(1e-10))
    x, y, z = P[..., 0:1], P[..., 1:2], P[..., 2:3]
    return np.concatenate([r, np.arctan2(y, x), np.arccos(np.clip(z / r, -1.0, 1.0))], axis=-1)


def spherical_to_cartesian(S):
    r, phi, theta = S[..., 0:1], S[..., 1:2], S[..., 2:3]
    return np.concatenate([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)], axis=-1)


class FrameState:
    """The four fields update_pointmap touches (frame.py:45-57): X_canon [N,3], C [N,1], N, N_updates."""

    def __init__(self, filtering_mode="weighted_pointmap", filtering_score="median"):
        self.X_canon = None
        self.C = None
        self.N = 0
        self.N_updates = 0
        self.mode = filtering_mode
        self.score_kind = filtering_score
        self._score = None

    def get_score(self, C):
        return float(np.median(C)) if self.score_kind == "median" else float(np.mean(C))

    def update_pointmap(self, X, C):
        X = np.asarray(X).reshape(-1, 3)
        C = np.asarray(C).reshape(-1, 1)
        if self.N == 0:
            self.X_canon, self.C, self.N, self.N_updates = X.copy(), C.copy(), 1, 1
            if self.mode == "best_score":
                self._score = self.get_score(C)
            return
        m = self.mode
        if m == "first":
            if self.N_updates == 1:
                self.X_canon, self.C, self.N = X.copy(), C.copy(), 1
        elif m == "recent":
            self.X_canon, self.C, self.N = X.copy(), C.copy(), 1
        elif m == "best_score":
            s = self.get_score(C)
            if s > (self._score or 0.0):
                self.X_canon, self.C, self.N, self._score = X.copy(), C.copy(), 1, s
        elif m == "indep_conf":
            new = C > self.C
            self.X_canon = np.where(new, X, self.X_canon)
            self.C = np.where(new, C, self.C)
            self.N = 1
        elif m == "weighted_pointmap":
            tot = self.C + C
            self.X_canon = (self.C * self.X_canon + C * X) / tot
            self.C = tot
            self.N += 1
        elif m == "weighted_spherical":
            tot = self.C + C
            sph = (self.C * cartesian_to_spherical(self.X_canon) + C * cartesian_to_spherical(X)) / tot
            self.X_canon = spherical_to_cartesian(sph)
            self.C = tot
            self.N += 1
        else:
            raise ValueError(m)
        self.N_updates += 1

    def get_average_conf(self):
        return None if self.C is None else self.C / self.N


def keyframe_stats(idx_f2k, valid_match, valid_kf):
    """tracker.py:149-155: (match_frac_k, unique_frac_f)."""
    n = valid_kf.size
    match_frac_k = float(np.sum(valid_kf.astype(np.float32))) / n
    unique_frac_f = np.unique(idx_f2k.reshape(-1)[valid_match.reshape(-1).astype(bool)]).shape[0] / n
    return match_frac_k, unique_frac_f
