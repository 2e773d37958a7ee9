"""CPU oracle (numpy, float64) for the backend Gauss-Newton "rays" solve.
TEST INFRASTRUCTURE ONLY.

Restates backends/mpsgraph/gauss_newton.py:23-280 (the numpy twin of
gn_jacobian_kernel, gauss_newton.metal:66-252): despite the name the residual
is the 3-D point error  Xj_Ci - Xi[idx]  (gauss_newton.py:144); sigma_dist is
accepted and ignored (:52).  PINNED against the importable reference twin via
tests/golden/gn_rays_*.npz.

Per directed edge e = (i, j) and point k passing
    valid & Q > Q_thresh & C_i[idx] > C_thresh & C_j > C_thresh        (:126-131)
  Tij = Ti^-1 Tj ; Y = sij R(qij) Xj + tij ; err = Y - Xi[idx]          (:118,:139-144)
  sqrt_w = sqrt(Q)/sigma ; w_c = huber(sqrt_w err_c) sqrt_w^2           (:147-152)
  base_c = [e_c | row c of -[Y]x | Y_c]                                 (:164-187)
  Jj_c = [R_i^T e_c / s_i | R_i^T base_rot_c | Y_c] ; Ji = -Jj          (:189-213)
  Hjj += sum_c w_c Jj_c Jj_c^T ; gj += sum_c w_c Jj_c err_c             (:230-240)
  (Ji = -Jj  =>  Hii = Hjj, Hij = -Hjj, gi = -gj)
then H += 1e-6 I, dx = solve(H, -g), stop if |dx| < delta_thresh (before the
update), T <- exp(dx) T for the free keyframes (:253-275).
"""
from __future__ import annotations

import numpy as np

from . import sim3 as S


def edge_blocks(t, q, s, Xs, Cs, ix, jx, idx_corr, valid, q_conf,
                sigma_ray=0.003, C_thresh=0.0, Q_thresh=1.5, point_mode=0, calib=None):
    """One edge -> (Hjj[7,7], gj[7], n_valid) in float64.  point_mode 0 rays / 1 points / 2 calib
    (calib = dict fx, fy, cx, cy, width, height, border, z_eps, sigma_pixel, sigma_depth)."""
    tij, qij, sij = S.sim3_relative(t[ix], q[ix], s[ix], t[jx], q[jx], s[jx])
    ci = Cs[ix, idx_corr]
    cj = Cs[jx]
    mask = valid & (q_conf > Q_thresh) & (ci > C_thresh) & (cj > C_thresh)
    vi = np.where(mask)[0]
    if len(vi) == 0:
        return np.zeros((7, 7)), np.zeros(7), 0
    Xi = Xs[ix, idx_corr[vi]].astype(np.float64)
    Xj = Xs[jx, vi].astype(np.float64)
    conf = q_conf[vi].astype(np.float64)
    Y = S.quat_rotate(qij[None], Xj) * sij + tij
    err = Y - Xi
    sqrt_w = (1.0 / sigma_ray) * np.sqrt(conf)
    if point_mode == 1:                  # gauss_newton_points.py:103-107
        sqrt_w = sqrt_w * (1.0 / (np.linalg.norm(Xi, axis=-1) + 1e-6))
    dproj = None
    if point_mode == 2:                  # gauss_newton_calib.py:118-190
        c = calib
        keep = (Y[:, 2] > c["z_eps"]) & (Xi[:, 2] > c["z_eps"])
        Xi, Y, conf = Xi[keep], Y[keep], conf[keep]
        if len(Xi) == 0:
            return np.zeros((7, 7)), np.zeros(7), 0
        zj, zi = 1.0 / Y[:, 2], 1.0 / Xi[:, 2]
        pju, pjv = c["fx"] * Y[:, 0] * zj + c["cx"], c["fy"] * Y[:, 1] * zj + c["cy"]
        piu, piv = c["fx"] * Xi[:, 0] * zi + c["cx"], c["fy"] * Xi[:, 1] * zi + c["cy"]
        inb = ((pju >= c["border"]) & (pju < c["width"] - c["border"]) & (pjv >= c["border"])
               & (pjv < c["height"] - c["border"]))
        if not inb.any():
            return np.zeros((7, 7)), np.zeros(7), 0
        Xi, Y, conf, zj = Xi[inb], Y[inb], conf[inb], zj[inb]
        isp, isd = 1.0 / c["sigma_pixel"], 1.0 / c["sigma_depth"]
        err = np.stack([(pju[inb] - piu[inb]) * isp, (pjv[inb] - piv[inb]) * isp,
                        (np.log(Y[:, 2]) - np.log(Xi[:, 2])) * isd], axis=-1)
        sqrt_w = np.sqrt(conf)
        dproj = np.zeros((len(Xi), 3, 3))
        dproj[:, 0, 0] = c["fx"] * zj * isp
        dproj[:, 0, 2] = -c["fx"] * Y[:, 0] * zj ** 2 * isp
        dproj[:, 1, 1] = c["fy"] * zj * isp
        dproj[:, 1, 2] = -c["fy"] * Y[:, 1] * zj ** 2 * isp
        dproj[:, 2, 2] = zj * isd
        vi = vi[:len(Xi)]                # only the count is used below
    w = S.huber_weight(sqrt_w[:, None] * err) * (sqrt_w[:, None] ** 2)      # [n,3]
    qi_inv = S.quat_inv(q[ix])
    s_inv = 1.0 / s[ix]
    n = len(vi)
    Jj = np.zeros((n, 3, 7))
    eye = np.eye(3)
    z = np.zeros(n)
    base_rot = np.stack([
        np.stack([z, Y[:, 2], -Y[:, 1]], -1),
        np.stack([-Y[:, 2], z, Y[:, 0]], -1),
        np.stack([Y[:, 1], -Y[:, 0], z], -1)], axis=1)                    # [n,3,3]
    for c in range(3):
        Jj[:, c, :3] = s_inv * S.quat_rotate(qi_inv, eye[c])[None, :]
        Jj[:, c, 3:6] = S.quat_rotate(qi_inv[None], base_rot[:, c, :])
        Jj[:, c, 6] = Y[:, c]
    if dproj is not None:
        Jj = np.einsum("nrc,nck->nrk", dproj, Jj)
    wJ = w[:, :, None] * Jj
    Hjj = np.einsum("nci,ncj->ij", Jj, wJ)
    gj = np.einsum("nci,nc->i", wJ, err)
    return Hjj, gj, n


def gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q,
                      sigma_ray=0.003, sigma_dist=10.0, C_thresh=0.0, Q_thresh=1.5,
                      max_iter=10, delta_thresh=1e-4, pin=1, return_info=False, point_mode=0, calib=None):
    """gauss_newton.py:23-280 (point_mode=True: gauss_newton_points.py:17-207).  Returns Twc_new [K,8] float32 (+ info dict)."""
    Twc = np.asarray(Twc)
    num_kf, num_edges = Twc.shape[0], len(ii)
    info = dict(iters=0, dx_norms=[], first_H=None, first_g=None)
    if num_edges == 0 or num_kf <= pin:
        return (Twc.copy(), info) if return_info else Twc.copy()
    unique_kf = np.unique(np.concatenate([ii, jj]))
    if len(unique_kf) <= pin:
        return (Twc.copy(), info) if return_info else Twc.copy()
    local = {int(kf): i - pin for i, kf in enumerate(unique_kf)}
    num_free = len(unique_kf) - pin
    valid_match = np.asarray(valid_match)
    Q = np.asarray(Q)
    Cs = np.asarray(Cs)
    if valid_match.ndim == 3:
        valid_match = valid_match[..., 0]
    if Q.ndim == 3:
        Q = Q[..., 0]
    if Cs.ndim == 3:
        Cs = Cs[..., 0]
    valid_match = valid_match.astype(bool)
    t = Twc[:, :3].astype(np.float64)
    q = Twc[:, 3:7].astype(np.float64)
    s = Twc[:, 7].astype(np.float64)
    dim = 7 * num_free
    for _ in range(max_iter):
        H = np.zeros((dim, dim))
        g = np.zeros(dim)
        for e in range(num_edges):
            ix, jx = int(ii[e]), int(jj[e])
            il, jl = local[ix], local[jx]
            if il < 0 and jl < 0:
                continue
            Hjj, gj, n = edge_blocks(t, q, s, Xs, Cs, ix, jx, idx_ii2jj[e], valid_match[e], Q[e],
                                     sigma_ray, C_thresh, Q_thresh, point_mode, calib)
            if n == 0:
                continue
            if il >= 0:
                H[il * 7:il * 7 + 7, il * 7:il * 7 + 7] += Hjj
                g[il * 7:il * 7 + 7] -= gj
            if jl >= 0:
                H[jl * 7:jl * 7 + 7, jl * 7:jl * 7 + 7] += Hjj
                g[jl * 7:jl * 7 + 7] += gj
            if il >= 0 and jl >= 0:
                H[il * 7:il * 7 + 7, jl * 7:jl * 7 + 7] -= Hjj
                H[jl * 7:jl * 7 + 7, il * 7:il * 7 + 7] -= Hjj.T
        H += np.eye(dim) * 1e-6
        if info["first_H"] is None:
            info["first_H"], info["first_g"] = H.copy(), g.copy()
        try:
            dx = np.linalg.solve(H, -g)
        except np.linalg.LinAlgError:
            break
        dn = float(np.linalg.norm(dx))
        info["dx_norms"].append(dn)
        if dn < delta_thresh:
            break
        info["iters"] += 1
        for i, kf in enumerate(unique_kf[pin:]):
            kf = int(kf)
            t[kf], q[kf], s[kf] = S.retract_sim3(dx[i * 7:i * 7 + 7], t[kf], q[kf], s[kf])
    out = np.concatenate([t, q, s[:, None]], axis=-1).astype(np.float32)
    return (out, info) if return_info else out


def cholesky_solve(H, g, reg=1e-6):
    """backends/mpsgraph/linalg.py:17-50: solve (H + reg I) x = g (LU), lstsq fallback."""
    H = np.asarray(H)
    g = np.asarray(g)
    g = g.squeeze() if g.ndim > 1 else g
    Hr = H + reg * np.eye(H.shape[0], dtype=H.dtype)
    try:
        return np.linalg.solve(Hr, g)
    except np.linalg.LinAlgError:
        return np.linalg.lstsq(Hr, g, rcond=None)[0]


def gauss_newton_points(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_point=0.01, C_thresh=0.0,
                        Q_thresh=1.5, max_iter=10, delta_thresh=1e-4, pin=1, return_info=False):
    """gauss_newton_points.py:17-207: rays variant + scale-invariant weight 1/(|Xi| + 1e-6)."""
    return gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_ray=sigma_point,
                             C_thresh=C_thresh, Q_thresh=Q_thresh, max_iter=max_iter, delta_thresh=delta_thresh,
                             pin=pin, return_info=return_info, point_mode=1)


def gauss_newton_calib(Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, img_size, pixel_border=0, z_eps=0.0,
                       sigma_pixel=1.0, sigma_depth=0.1, C_thresh=0.0, Q_thresh=1.5, max_iter=10,
                       delta_thresh=1e-4, pin=1, return_info=False):
    """gauss_newton_calib.py:17-274: pixel + log-depth residual; img_size = (width, height)."""
    K = np.asarray(K)
    fx, fy, cx, cy = (K[0, 0], K[1, 1], K[0, 2], K[1, 2]) if K.shape == (3, 3) else K.flatten()[:4]
    calib = dict(fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy), width=img_size[0], height=img_size[1],
                 border=pixel_border, z_eps=z_eps, sigma_pixel=sigma_pixel, sigma_depth=sigma_depth)
    return gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_ray=1.0, C_thresh=C_thresh,
                             Q_thresh=Q_thresh, max_iter=max_iter, delta_thresh=delta_thresh, pin=pin,
                             return_info=return_info, point_mode=2, calib=calib)
