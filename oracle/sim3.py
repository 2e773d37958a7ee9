"""CPU oracle (numpy) for quaternion / Sim(3) algebra.  TEST INFRASTRUCTURE ONLY.

Two families exist in the reference and both are restated here:

  * "backend" ops used by the numpy Gauss-Newton twins
    (backends/mpsgraph/sim3_ops.py): quat_multiply :16, quat_inv :39,
    quat_rotate :51, sim3_act :75, sim3_relative :96, exp_so3 :129,
    exp_sim3 :161 (full W matrix), retract_sim3 :229 (LEFT multiply),
    huber_weight :295.  EPS = 1e-6 there.
  * "tracker" ops of the MLX Lie-group classes (liegroups/so3.py, sim3.py):
    SO3.exp so3.py:65-96, SO3.act :157-172, Sim3.exp sim3.py:107-154
    (SE3-style V, no scale coupling), Sim3.inv :195-204 (1/(s+1e-10)),
    Sim3.__mul__ :206-220, Sim3.act :222-231, Sim3.retr :253-262 (RIGHT multiply).

Pose layout everywhere: [tx,ty,tz, qx,qy,qz,qw, s].  dtype follows the input
(float64 in the oracle's default use).
"""
from __future__ import annotations

import numpy as np

EPS = 1e-6


# ---------------------------------------------------------------- quaternions
def quat_multiply(q1, q2):
    x1, y1, z1, w1 = q1[..., 0], q1[..., 1], q1[..., 2], q1[..., 3]
    x2, y2, z2, w2 = q2[..., 0], q2[..., 1], q2[..., 2], q2[..., 3]
    return np.stack([
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
    ], axis=-1)


def quat_inv(q):
    return np.stack([-q[..., 0], -q[..., 1], -q[..., 2], q[..., 3]], axis=-1)


def quat_rotate(q, v):
    """v + qw*(2 q x v) + q x (2 q x v)   (sim3_ops.py:51-72 == so3.py:157-172)."""
    qx, qy, qz, qw = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    vx, vy, vz = v[..., 0], v[..., 1], v[..., 2]
    ux = 2.0 * (qy * vz - qz * vy)
    uy = 2.0 * (qz * vx - qx * vz)
    uz = 2.0 * (qx * vy - qy * vx)
    return np.stack([
        vx + qw * ux + (qy * uz - qz * uy),
        vy + qw * uy + (qz * ux - qx * uz),
        vz + qw * uz + (qx * uy - qy * ux),
    ], axis=-1)


def cross(a, b):
    return np.stack([
        a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0],
    ], axis=-1)


def huber_weight(r, k=1.345):
    """sim3_ops.py:295-306 == optimizer.py:49-62: 1 if |r|<k else k/|r|."""
    ra = np.abs(r)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(ra < k, 1.0, k / ra)


# ---------------------------------------------------------------- backend family
def sim3_act(t, q, s, X):
    return quat_rotate(q, X) * np.asarray(s)[..., None] + t


def sim3_relative(ti, qi, si, tj, qj, sj):
    """Tij = Ti^-1 * Tj  (sim3_ops.py:96-126)."""
    si_inv = 1.0 / si
    sij = si_inv * sj
    qi_inv = quat_inv(qi)
    qij = quat_multiply(qi_inv, qj)
    tij = quat_rotate(qi_inv, tj - ti) * np.asarray(si_inv)[..., None]
    return tij, qij, sij


def exp_so3(phi):
    """sim3_ops.py:129-158."""
    th2 = np.sum(phi * phi, axis=-1)
    th = np.sqrt(th2 + EPS)
    small = th2 < EPS
    imag = np.where(small, 0.5 - th2 / 48.0, np.sin(0.5 * th) / th)
    real = np.where(small, 1.0 - th2 / 8.0, np.cos(0.5 * th))
    return np.stack([imag * phi[..., 0], imag * phi[..., 1], imag * phi[..., 2], real], axis=-1)


def exp_sim3(xi):
    """sim3_ops.py:161-226, full W = C I + A [w]x + B [w]x^2."""
    tau, omega, sigma = xi[..., :3], xi[..., 3:6], xi[..., 6]
    q = exp_so3(omega)
    s = np.exp(sigma)
    th2 = np.sum(omega * omega, axis=-1)
    th = np.sqrt(th2 + EPS)
    small_t = th2 < EPS
    small_s = np.abs(sigma) < EPS
    with np.errstate(divide="ignore", invalid="ignore"):
        C = np.where(small_s, 1.0, (s - 1.0) / sigma)
        A = np.where(
            small_s,
            np.where(small_t, 0.5, (1.0 - np.cos(th)) / th2),
            np.where(small_t, ((sigma - 1.0) * s + 1.0) / (sigma * sigma),
                     (s * np.sin(th) * sigma + (1.0 - s * np.cos(th)) * th)
                     / (th * (th2 + sigma * sigma))))
        B = np.where(
            small_s,
            np.where(small_t, 1.0 / 6.0, (th - np.sin(th)) / (th2 * th)),
            np.where(small_t, (s * 0.5 * sigma * sigma + s - 1.0 - sigma * s) / (sigma * sigma * sigma),
                     (C - ((s * np.cos(th) - 1.0) * sigma + s * np.sin(th) * th)
                      / (th2 + sigma * sigma)) / th2))
    c1 = cross(omega, tau)
    c2 = cross(omega, c1)
    t = np.asarray(C)[..., None] * tau + np.asarray(A)[..., None] * c1 + np.asarray(B)[..., None] * c2
    return t, q, s


def retract_sim3(xi, t, q, s):
    """T_new = exp(xi) * T  (sim3_ops.py:229-251)."""
    dt, dq, ds = exp_sim3(xi)
    return quat_rotate(dq, t) * np.asarray(ds)[..., None] + dt, quat_multiply(dq, q), ds * s


# ---------------------------------------------------------------- tracker family (pose = [..., 8])
def so3_exp_mlx(omega):
    """so3.py:65-96 (eps 1e-10 inside the sqrt, small-angle switch at theta^2 < 1e-8)."""
    th2 = np.sum(omega * omega, axis=-1, keepdims=True)
    th = np.sqrt(th2 + 1e-10)
    small = th2 < 1e-8
    sinc_half = np.where(small, 0.5 - th2 / 48.0, np.sin(0.5 * th) / th)
    cos_half = np.where(small, 1.0 - th2 / 8.0, np.cos(0.5 * th))
    return np.concatenate([sinc_half * omega, cos_half], axis=-1)


def _safe_div(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return a / b


def sim3_exp_mlx(tau):
    """sim3.py:107-154: t = v + B w x v + C w x (w x v), s = exp(sigma)."""
    v, omega, sigma = tau[..., :3], tau[..., 3:6], tau[..., 6:7]
    th2 = np.sum(omega * omega, axis=-1, keepdims=True)
    th = np.sqrt(th2 + 1e-10)
    small = th2 < 1e-8
    s = np.exp(sigma)
    A = np.where(small, 1.0 - th2 / 6.0, np.sin(th) / th)
    B = np.where(small, 0.5 - th2 / 24.0, _safe_div(1.0 - np.cos(th), th2))
    wv = cross(omega, v)
    C = np.where(small, 1.0 / 6.0 - th2 / 120.0, _safe_div(1.0 - A, th2))
    t = v + B * wv + C * cross(omega, wv)
    return np.concatenate([t, so3_exp_mlx(omega), s], axis=-1)


def sim3_inv_mlx(T):
    """sim3.py:195-204."""
    t, q, s = T[..., :3], T[..., 3:7], T[..., 7:8]
    qi = quat_inv(q)
    si = 1.0 / (s + 1e-10)
    return np.concatenate([-si * quat_rotate(qi, t), qi, si], axis=-1)


def sim3_mul_mlx(T1, T2):
    """sim3.py:206-220."""
    t1, q1, s1 = T1[..., :3], T1[..., 3:7], T1[..., 7:8]
    t2, q2, s2 = T2[..., :3], T2[..., 3:7], T2[..., 7:8]
    return np.concatenate([t1 + s1 * quat_rotate(q1, t2), quat_multiply(q1, q2), s1 * s2], axis=-1)


def sim3_act_mlx(T, p):
    """sim3.py:222-231: s R p + t."""
    return T[..., 7:8] * quat_rotate(T[..., 3:7], p) + T[..., :3]


def sim3_retr_mlx(T, tau):
    """sim3.py:253-262: T * exp(tau)."""
    return sim3_mul_mlx(T, sim3_exp_mlx(tau))
