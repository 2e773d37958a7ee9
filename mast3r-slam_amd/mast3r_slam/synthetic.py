"""Seeded synthetic inputs for tests and benchmarks (numpy only, no reference code).

Recipes follow SURVEY.md §8d:
  * textured_image      low-frequency uint8 texture for the network
  * geometric_pair      smooth surface seen by two views (pointmaps in view-1's
                        frame), 24-d sinusoidal descriptors, known true matches
  * tracking_problem    config 3: frame points = T^-1 * keyframe points + noise
  * gn_graph            create_gn_test_data-style random pose graph
                        (shape/dtype recipe of benchmark_all_kernels.py:16-42,
                        own generator and RNG stream)
  * keyframe_graph_scene  config 5: K keyframes on a circular trajectory viewing the same
                        smooth surface - canonical pointmaps, world points, descriptors,
                        poses and the true pixel correspondences of any edge (torch, on the
                        device: 256 keyframes x 262144 points are generated where they are used)
"""
from __future__ import annotations

import numpy as np


def textured_image(h: int, w: int, seed: int) -> np.ndarray:
    """Bilinear-upsampled 12x16 uniform[0,255] grid + N(0,4) noise -> uint8 [h,w,3]."""
    rng = np.random.default_rng(seed)
    gh, gw = 12, 16
    grid = rng.uniform(0, 255, size=(gh, gw, 3))
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.clip(np.floor(ys).astype(int), 0, gh - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, gw - 2)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    img = ((1 - fy) * (1 - fx) * grid[y0][:, x0] + (1 - fy) * fx * grid[y0][:, x0 + 1]
           + fy * (1 - fx) * grid[y0 + 1][:, x0] + fy * fx * grid[y0 + 1][:, x0 + 1])
    img = img + rng.normal(0, 4, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _surface(u, v, h, w):
    """Continuous surface point for (possibly fractional) pixel (u, v), f = w."""
    z = 2.0 + 0.3 * np.sin(2 * np.pi * u / w) * np.cos(2 * np.pi * v / h)
    return np.stack([(u - 0.5 * w) / w * z, (v - 0.5 * h) / w * z, z], axis=-1)


def _descriptor(X, d=24):
    """d-dim sinusoidal embedding of 3-D points, L2-normalised (float32)."""
    k = d // 6
    freqs = 2.0 ** np.arange(k) * 3.0
    ang = X[..., None, :] * freqs[:, None]                       # [...,k,3]
    e = np.concatenate([np.sin(ang), np.cos(ang)], axis=-1).reshape(X.shape[:-1] + (-1,))
    e = e[..., :d]
    return (e / np.linalg.norm(e, axis=-1, keepdims=True)).astype(np.float32)


def geometric_pair(h: int, w: int, seed: int = 0, batch: int = 1, noise: float = 1e-4, d: int = 24):
    """A smooth two-view scene.

    Returns dict with X11,X21 [B,h,w,3] f32 (both in view-1's frame), D11,D21
    [B,h,w,d] f32, and uv_true [B,h*w,2] f32: where view-2 pixel n truly lands in
    view-1's image (so iter_proj should converge to it).
    """
    rng = np.random.default_rng(seed)
    vv, uu = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    out = {k: [] for k in ("X11", "X21", "D11", "D21", "uv_true")}
    for _ in range(batch):
        a = rng.uniform(-1, 1, size=6)
        # smooth warp view-2 pixel -> view-1 location (a few pixels, slowly varying)
        u1 = uu + 4.3 * a[0] + 2.0 * a[1] * np.sin(2 * np.pi * vv / h + a[2])
        v1 = vv + 3.1 * a[3] + 1.5 * a[4] * np.cos(2 * np.pi * uu / w + a[5])
        X11 = _surface(uu, vv, h, w)
        X21 = _surface(u1, v1, h, w)
        X11n = X11 + rng.normal(0, noise, X11.shape)
        X21n = X21 + rng.normal(0, noise, X21.shape)
        out["X11"].append(X11n.astype(np.float32))
        out["X21"].append(X21n.astype(np.float32))
        out["D11"].append(_descriptor(X11, d))
        out["D21"].append(_descriptor(X21, d))
        out["uv_true"].append(np.stack([u1, v1], -1).reshape(h * w, 2).astype(np.float32))
    return {k: np.stack(v) for k, v in out.items()}


def _quat_rot(q, v):
    u = 2.0 * np.cross(q[:3], v)
    return v + q[3] * u + np.cross(q[:3], u)


def tracking_problem(h: int, w: int, seed: int = 0, noise: float = 1e-3, perm_frac: float = 0.05,
                     valid_p: float = 0.7):
    """SURVEY §8d config 3.  Returns dict: Xf_canon [N,3] (ungathered frame points),
    Xk [N,3], idx [N] int64, Qk [N] f32, valid [N] bool, T_WCf, T_WCk [8] f32, T_true [8]
    (the T_CkCf that maps frame points onto keyframe points)."""
    rng = np.random.default_rng(seed)
    n = h * w
    vv, uu = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    Xk = _surface(uu, vv, h, w).reshape(n, 3)
    ang = np.deg2rad(2.0)
    q = np.array([0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)])
    t = np.array([0.05, 0.0, 0.01])
    s = 1.02
    # Xk = s R Xf + t  ->  Xf = R^T (Xk - t) / s
    qi = np.array([-q[0], -q[1], -q[2], q[3]])
    Xf_true = np.stack([_quat_rot(qi, x) for x in (Xk - t)]) / s if n <= 4096 else None
    if Xf_true is None:                                   # vectorised for big N
        v = Xk - t
        u = 2.0 * np.cross(np.broadcast_to(qi[:3], v.shape), v)
        Xf_true = (v + qi[3] * u + np.cross(np.broadcast_to(qi[:3], v.shape), u)) / s
    Xf_g = Xf_true + rng.normal(0, noise, Xf_true.shape)  # gathered order (matches Xk row by row)
    idx = np.arange(n, dtype=np.int64)
    m = int(perm_frac * n)
    if m > 1:
        sel = rng.choice(n, size=m, replace=False)
        idx[sel] = sel[rng.permutation(m)]
    # canonical frame map such that Xf_canon[idx] == Xf_g wherever idx is a bijection
    Xf_canon = np.zeros_like(Xf_g)
    Xf_canon[idx] = Xf_g
    Qk = rng.uniform(1.0, 4.0, size=n).astype(np.float32)
    valid = rng.uniform(size=n) < valid_p
    ident = np.array([0, 0, 0, 0, 0, 0, 1, 1], dtype=np.float32)
    return dict(Xf_canon=Xf_canon.astype(np.float32), Xk=Xk.astype(np.float32), idx=idx, Qk=Qk,
                valid=valid, T_WCf=ident.copy(), T_WCk=ident.copy(),
                T_true=np.concatenate([t, q, [s]]).astype(np.float32))


def gn_graph(num_kf=10, num_pts=500, num_edges=15, seed=42, chain=False, pose_noise=0.0, poses=None):
    """Random pose graph with the array layout of benchmark_all_kernels.py:16-42.

    chain=True: edges connect each keyframe to its previous <=3 (slam.py:302-303) and the
    points are one shared cloud seen from perturbed poses (a solvable problem)."""
    rng = np.random.default_rng(seed)
    Twc = np.zeros((num_kf, 8), dtype=np.float32)
    for i in range(num_kf):
        Twc[i, :3] = rng.normal(size=3) * 0.5
        q = rng.normal(size=4)
        Twc[i, 3:7] = q / np.linalg.norm(q)
        Twc[i, 7] = 1.0 + rng.uniform() * 0.1
    if poses is not None:
        Twc = np.asarray(poses, dtype=np.float32).copy()
    if not chain:
        Xs = rng.normal(size=(num_kf, num_pts, 3)).astype(np.float32)
        ii = rng.integers(0, num_kf, num_edges).astype(np.int32)
        jj = rng.integers(0, num_kf, num_edges).astype(np.int32)
        for e in range(num_edges):
            while jj[e] == ii[e]:
                jj[e] = rng.integers(0, num_kf)
        idx = np.stack([rng.permutation(num_pts) for _ in range(num_edges)]).astype(np.int32)
    else:
        world = rng.normal(size=(num_pts, 3)) + np.array([0, 0, 4.0])
        Xs = np.zeros((num_kf, num_pts, 3), dtype=np.float32)
        for i in range(num_kf):
            t, q, s = Twc[i, :3].astype(np.float64), Twc[i, 3:7].astype(np.float64), float(Twc[i, 7])
            qi = np.array([-q[0], -q[1], -q[2], q[3]])
            v = world - t
            u = 2.0 * np.cross(np.broadcast_to(qi[:3], v.shape), v)
            Xs[i] = ((v + qi[3] * u + np.cross(np.broadcast_to(qi[:3], v.shape), u)) / s).astype(np.float32)
        pairs = [(i, j) for i in range(num_kf) for j in range(max(0, i - 3), i)]
        ii = np.array([p[0] for p in pairs] + [p[1] for p in pairs], dtype=np.int32)
        jj = np.array([p[1] for p in pairs] + [p[0] for p in pairs], dtype=np.int32)
        num_edges = len(ii)
        idx = np.broadcast_to(np.arange(num_pts, dtype=np.int32), (num_edges, num_pts)).copy()
        if pose_noise > 0:
            Twc[1:, :3] += rng.normal(size=(num_kf - 1, 3)).astype(np.float32) * pose_noise
    Cs = (rng.uniform(size=(num_kf, num_pts)) * 10 + 1).astype(np.float32)
    valid = rng.uniform(size=(num_edges, num_pts)) > 0.3
    Q = (rng.uniform(size=(num_edges, num_pts)) * 3 + 1).astype(np.float32)
    return Twc, Xs, Cs, ii, jj, idx, valid, Q


def chain_edges(num_kf: int, back: int = 3):
    """slam.py:302-303: every keyframe is linked to its previous <= `back` keyframes.  Returns (ii, jj) lists with
    ii < jj (256 keyframes -> 762 undirected edges)."""
    ii, jj = [], []
    for j in range(1, num_kf):
        for i in range(max(0, j - back), j):
            ii.append(i)
            jj.append(j)
    return ii, jj


def keyframe_graph_scene(num_kf: int, h: int, w: int, device, seed: int = 0, d: int = 24, radius_px: float = 40.0,
                         desc_dtype=None):
    """`num_kf` keyframes on a circular trajectory looking at one smooth surface.

    Keyframe k's pixel (u, v) sees the surface point S(u + a_k, v + b_k) with (a_k, b_k) = radius_px (cos, sin)(2 pi k / K)
    and has the world pose T_k = (t_k, q_k, s_k) (small translation on the circle, a rotation about y that varies along
    it, a scale near one).  Hence for an edge (i, j) the TRUE match of keyframe j's pixel (u, v) in keyframe i is
    (u + a_j - a_i, v + b_j - b_i), and T_i X_i[match] = T_j X_j up to interpolation.

    Returns a dict of torch tensors on `device`:
      Pw [K,N,3] f32 world points, Xs [K,N,3] f32 canonical pointmaps (T_k^-1 Pw_k), D [K,H,W,d] descriptors of the
      world points (desc_dtype, default float16), poses [K,8] f32 (t, q_xyzw, s), shift [K,2] f64 (a_k, b_k),
      C [K,N] f32 confidences U(1,3), Qself / Qother [K,N] f32 descriptor confidences U(1,4)."""
    import torch
    dd = desc_dtype or torch.float16
    f64 = torch.float64
    g = torch.Generator(device="cpu").manual_seed(seed)
    k = torch.arange(num_kf, dtype=f64, device=device)
    th = 2 * np.pi * k / num_kf
    shift = torch.stack([radius_px * torch.cos(th), radius_px * torch.sin(th)], -1)              # [K,2]
    t = torch.stack([0.05 * torch.cos(th), 0.05 * torch.sin(th), 0.01 * torch.sin(2 * th)], -1)  # [K,3]
    ang = np.deg2rad(2.0) * torch.sin(th)
    q = torch.stack([torch.zeros_like(ang), torch.sin(ang / 2), torch.zeros_like(ang), torch.cos(ang / 2)], -1)
    sc = 1.0 + 0.02 * torch.cos(th)
    n = h * w
    vv, uu = torch.meshgrid(torch.arange(h, dtype=f64, device=device), torch.arange(w, dtype=f64, device=device), indexing="ij")
    Pw = torch.empty((num_kf, n, 3), dtype=torch.float32, device=device)
    Xs = torch.empty((num_kf, n, 3), dtype=torch.float32, device=device)
    D = torch.empty((num_kf, h, w, d), dtype=dd, device=device)
    kk = d // 6
    freqs = (2.0 ** torch.arange(kk, dtype=f64, device=device)) * 3.0
    for i in range(num_kf):
        u = uu + shift[i, 0]
        v = vv + shift[i, 1]
        z = 2.0 + 0.3 * torch.sin(2 * np.pi * u / w) * torch.cos(2 * np.pi * v / h)
        P = torch.stack([(u - 0.5 * w) / w * z, (v - 0.5 * h) / w * z, z], -1).reshape(n, 3)     # _surface
        Pw[i] = P.float()
        # X = R^T (P - t) / s with R = rotation about y by ang
        c, s_ = torch.cos(ang[i]), torch.sin(ang[i])
        pv = P - t[i]
        X = torch.stack([c * pv[:, 0] - s_ * pv[:, 2], pv[:, 1], s_ * pv[:, 0] + c * pv[:, 2]], -1) / sc[i]
        Xs[i] = X.float()
        a = P[:, None, :] * freqs[:, None]                                                      # _descriptor
        e = torch.cat([torch.sin(a), torch.cos(a)], -1).reshape(n, -1)[:, :d]
        D[i] = (e / e.norm(dim=-1, keepdim=True)).reshape(h, w, d).to(dd)
    poses = torch.cat([t, q, sc[:, None]], -1).float()
    rnd = lambda lo, hi: (torch.rand((num_kf, n), generator=g) * (hi - lo) + lo).to(device)
    return dict(Pw=Pw, Xs=Xs, D=D, poses=poses, shift=shift, C=rnd(1.0, 3.0), Qself=rnd(1.0, 4.0), Qother=rnd(1.0, 4.0))
