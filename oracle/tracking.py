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


# ------------------------------------------------------------------ calibrated tracking (tracker.py:326-406)
def project_calib(P, K, img_size, jacobian=False, border=0, z_eps=0.0):
    """geometry.py:156-227.  P [...,3], K [3,3], img_size = (height, width).
    Returns (pz [...,3] = (u, v, log z), valid [...,1]) or (pz, dpz_dP [...,3,3], valid)."""
    h, w = img_size
    p = np.einsum("ij,...j->...i", K, P)
    z = p[..., 2:3]
    uv = (p / (z + 1e-10))[..., :2]
    u, v = uv[..., 0:1], uv[..., 1:2]
    valid_z = P[..., 2:3] > z_eps
    valid = (u > border) & (u < w - 1 - border) & (v > border) & (v < h - 1 - border) & valid_z
    with np.errstate(invalid="ignore", divide="ignore"):
        logz = np.where(valid_z, np.log(P[..., 2:3] + 1e-10), 0.0)
    pz = np.concatenate([uv, logz], axis=-1)
    if not jacobian:
        return pz, valid
    fx, fy = K[0, 0], K[1, 1]
    x, y, zp = P[..., 0], P[..., 1], P[..., 2]
    zi = 1.0 / (zp + 1e-10)
    J = np.zeros(P.shape[:-1] + (3, 3), dtype=P.dtype)
    J[..., 0, 0] = fx * zi
    J[..., 0, 2] = -fx * x * zi * zi
    J[..., 1, 1] = fy * zi
    J[..., 1, 2] = -fy * y * zi * zi
    J[..., 2, 2] = zi
    return pz, J, valid


def constrain_points_to_ray(img_size, X, K):
    """geometry.py:273-302 + backproject :229-245: keep z, put the point on its pixel's ray.  X [H*W,3]."""
    h, w = img_size
    vv, uu = np.meshgrid(np.arange(h, dtype=X.dtype), np.arange(w, dtype=X.dtype), indexing="ij")
    u, v = uu.reshape(-1), vv.reshape(-1)
    z = X[..., 2]
    return np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=-1)


def calib_measurements(Xk, img_size, depth_eps=0.0):
    """tracker.py:203-212: meas_k = (u, v, log(z + 1e-10)) zeroed where z <= depth_eps; valid_meas_k."""
    h, w = img_size
    vv, uu = np.meshgrid(np.arange(h, dtype=Xk.dtype), np.arange(w, dtype=Xk.dtype), indexing="ij")
    valid = Xk[..., 2:3] > depth_eps
    with np.errstate(invalid="ignore", divide="ignore"):
        meas = np.concatenate([uu.reshape(-1, 1), vv.reshape(-1, 1), np.log(Xk[..., 2:3] + 1e-10)], axis=-1)
    return np.where(valid, meas, 0.0), valid


def opt_pose_calib_sim3(Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, cfg=None, dtype=np.float64, fixed_iters=None):
    """tracker.py:326-406.  Xf [N,3] (ray-constrained frame points gathered at idx_f2k), Xk [N,3]
    (ray-constrained keyframe points, pixel order), K [3,3], img_size = (height, width)."""
    c = dict(DEFAULT_CFG, sigma_pixel=1.0, sigma_depth=10.0, pixel_border=0, depth_eps=0.0)
    c.update(cfg or {})
    Xf = np.asarray(Xf, dtype=dtype).reshape(-1, 3)
    Xk = np.asarray(Xk, dtype=dtype).reshape(-1, 3)
    K = np.asarray(K, dtype=dtype)
    Qk = np.asarray(Qk, dtype=dtype).reshape(-1, 1)
    v = np.asarray(valid).reshape(-1, 1).astype(dtype)
    T_WCf = np.asarray(T_WCf, dtype=dtype).reshape(8)
    T_WCk = np.asarray(T_WCk, dtype=dtype).reshape(8)
    meas_k, valid_meas = calib_measurements(Xk, img_size, c["depth_eps"])
    si_px = dtype(1.0 / c["sigma_pixel"]) * v * np.sqrt(Qk)
    si_d = dtype(1.0 / c["sigma_depth"]) * v * np.sqrt(Qk)
    sqrt_info = np.concatenate([np.broadcast_to(si_px, si_px.shape[:-1] + (2,)), si_d], axis=-1)
    T = S.sim3_mul_mlx(S.sim3_inv_mlx(T_WCk), T_WCf)
    old_cost = float("inf")
    n_it = c["max_iters"] if fixed_iters is None else fixed_iters
    costs = []
    for _ in range(n_it):
        Xf_Ck, dX_dT = act_sim3(T, Xf, jacobian=True)
        pz, dpz, valid_proj = project_calib(Xf_Ck, K, img_size, True, c["pixel_border"], c["depth_eps"])
        si2 = np.where(valid_proj & valid_meas, sqrt_info, 0.0)
        r = meas_k - pz
        J = -dpz @ dX_dT
        tau, new_cost, _, _ = solve_step(si2, r, J, c["huber"])
        T = S.sim3_retr_mlx(T, tau.astype(dtype))
        costs.append(new_cost)
        if fixed_iters is None and check_convergence(c["rel_error"], c["delta_norm"], old_cost, new_cost, tau):
            break
        old_cost = new_cost
    return S.sim3_mul_mlx(T_WCk, T), T, dict(iters=len(costs), costs=costs)
