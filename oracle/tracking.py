"""CPU oracle (numpy) for the frame-to-keyframe Gauss-Newton tracking solve.
TEST INFRASTRUCTURE ONLY.

Restated from MLX source text (not executable here - `mlx` is absent):
  * FrameTracker._opt_pose_ray_dist_sim3   tracker.py:258-324
  * FrameTracker._solve                    tracker.py:216-256
  * Qk / validity masks in track()         tracker.py:88-113
  * act_Sim3, point_to_ray_dist            geometry.py:34-137
  * check_convergence, huber_weight        optimizer.py:11-62
  * cholesky_solve (H + reg I, LU solve)   backends/mpsgraph/linalg.py:17-50
Parity status: UNPINNED against reference outputs (no executable reference);
pinned by analytic tests (finite-difference Jacobians, recover-a-known-Sim3).

The reference runs this in float32 MLX; the oracle defaults to float64 and is
the accuracy yardstick for the float32 HIP path (tolerance stated in tests).
Note the reference's quirk, reproduced: the Jacobian is the LEFT-perturbation
one ([I, -[p]x, p], geometry.py:118-137) while the retraction multiplies on the
RIGHT (sim3.py:253-262).
"""
from __future__ import annotations

import numpy as np

from . import sim3 as S

DEFAULT_CFG = dict(max_iters=10, huber=1.345, sigma_ray=0.003, sigma_dist=10.0,
                   C_conf=0.0, Q_conf=1.5, rel_error=1e-3, delta_norm=1e-3,
                   min_match_frac=0.05)


def point_to_ray_dist(X, jacobian=False):
    """geometry.py:46-92.  rd = [X/d, d], d = sqrt(|X|^2 + 1e-10)."""
    d = np.sqrt(np.sum(X * X, axis=-1, keepdims=True) + 1e-10)
    d_inv = 1.0 / d
    r = d_inv * X
    rd = np.concatenate([r, d], axis=-1)
    if not jacobian:
        return rd
    eye = np.eye(3, dtype=X.dtype)
    dr = d_inv[..., None] * (eye - (d_inv ** 2)[..., None] * (X[..., :, None] * X[..., None, :]))
    return rd, np.concatenate([dr, r[..., None, :]], axis=-2)


def skew(x):
    z = np.zeros_like(x[..., 0])
    return np.stack([
        np.stack([z, -x[..., 2], x[..., 1]], -1),
        np.stack([x[..., 2], z, -x[..., 0]], -1),
        np.stack([-x[..., 1], x[..., 0], z], -1)], -2)


def act_sim3(T, p, jacobian=False):
    """geometry.py:95-137: pW = s R p + t ; J = [I, -[pW]x, pW]."""
    pW = S.sim3_act_mlx(T, p)
    if not jacobian:
        return pW
    eye = np.broadcast_to(np.eye(3, dtype=pW.dtype), pW.shape[:-1] + (3, 3))
    return pW, np.concatenate([eye, -skew(pW), pW[..., None]], axis=-1)


def match_quality(Qff, Qkf, idx_f2k):
    """tracker.py:88-91: Qk = sqrt(Qff[idx] * Qkf)."""
    return np.sqrt(Qff[idx_f2k] * Qkf)


def validity(valid_match, Cf, Ck, Qk, C_conf=0.0, Q_conf=1.5):
    """tracker.py:108-113 -> (valid_opt, valid_kf)."""
    vq = Qk > Q_conf
    return valid_match & (Cf > C_conf) & (Ck > C_conf) & vq, valid_match & vq


def solve_step(sqrt_info, r, J, huber_k=1.345, reg=1e-6):
    """tracker.py:216-256 -> (tau[7], cost, H, g)."""
    wr = sqrt_info * r
    rsi = sqrt_info * np.sqrt(S.huber_weight(wr, huber_k))
    A = (rsi[..., None] * J).reshape(-1, J.shape[-1])
    b = (rsi * r).reshape(-1, 1)
    H = A.T @ A
    g = -(A.T @ b)[:, 0]
    cost = 0.5 * float((b.T @ b)[0, 0])
    tau = np.linalg.solve(H + reg * np.eye(H.shape[0], dtype=H.dtype), g)
    return tau, cost, H, g


def check_convergence(rel_thr, dn_thr, old_cost, new_cost, tau):
    """optimizer.py:11-46 (nan at step 0 from inf/inf compares False, as in the reference)."""
    with np.errstate(invalid="ignore"):
        rel_dec = abs(np.float64(old_cost - new_cost) / np.float64(old_cost + 1e-10))
    dn = float(np.sqrt(np.sum(tau * tau)))
    return bool(rel_dec < rel_thr) or dn < dn_thr


def opt_pose_ray_dist_sim3(Xf, Xk, T_WCf, T_WCk, Qk, valid, cfg=None, dtype=np.float64,
                           fixed_iters=None):
    """tracker.py:258-324.

    Xf [N,3] frame points already gathered at idx_f2k, Xk [N,3], poses [8],
    Qk [N] or [N,1], valid [N] or [N,1].  Returns (T_WCf[8], T_CkCf[8], info).
    fixed_iters: run exactly that many iterations (convergence test disabled) -
    used for timing-equivalent parity.
    """
    c = dict(DEFAULT_CFG)
    c.update(cfg or {})
    Xf = np.asarray(Xf, dtype=dtype).reshape(-1, 3)
    Xk = np.asarray(Xk, dtype=dtype).reshape(-1, 3)
    Qk = np.asarray(Qk, dtype=dtype).reshape(-1, 1)
    v = np.asarray(valid).reshape(-1, 1).astype(dtype)
    T_WCf = np.asarray(T_WCf, dtype=dtype).reshape(8)
    T_WCk = np.asarray(T_WCk, dtype=dtype).reshape(8)
    si_ray = dtype(1.0 / c["sigma_ray"]) * v * np.sqrt(Qk)
    si_dist = dtype(1.0 / c["sigma_dist"]) * v * np.sqrt(Qk)
    sqrt_info = np.concatenate([np.broadcast_to(si_ray, si_ray.shape[:-1] + (3,)), si_dist], axis=-1)
    T = S.sim3_mul_mlx(S.sim3_inv_mlx(T_WCk), T_WCf)
    rd_k = point_to_ray_dist(Xk)
    old_cost = float("inf")
    n_it = c["max_iters"] if fixed_iters is None else fixed_iters
    costs, taus = [], []
    for _ in range(n_it):
        Xf_Ck, dX_dT = act_sim3(T, Xf, jacobian=True)
        rd_f, drd_dX = point_to_ray_dist(Xf_Ck, jacobian=True)
        r = rd_k - rd_f
        J = -drd_dX @ dX_dT
        tau, new_cost, _, _ = solve_step(sqrt_info, r, J, c["huber"])
        tau = tau.astype(dtype)
        T = S.sim3_retr_mlx(T, tau)
        costs.append(new_cost)
        taus.append(tau)
        if fixed_iters is None and check_convergence(c["rel_error"], c["delta_norm"], old_cost, new_cost, tau):
            break
        old_cost = new_cost
    return S.sim3_mul_mlx(T_WCk, T), T, dict(iters=len(costs), costs=costs, taus=taus)
