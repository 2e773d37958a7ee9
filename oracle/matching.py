"""CPU oracle (numpy) for the dense-matching stage.  TEST INFRASTRUCTURE ONLY.

Restates, with a fully pinned floating-point operation order, the reference's
  * prep_for_iter_proj / normalize_rays   matching.py:121-175, image.py:9-34
  * _iter_proj_numpy                      backends/mpsgraph/kernels.py:151-254
  * _refine_matches_numpy                 backends/mpsgraph/kernels.py:496-537
  * match_iterative_proj epilogue         matching.py:436-461
  * match_simple                          matching.py:41-90
(paths relative to /root/reference/src/mlx_mast3r_slam).

The HIP kernels implement exactly the operation order written here (every
float32 op individually rounded, no FMA contraction), so GPU-vs-oracle parity
is bit-exact on p / valid / idx.  Where the reference's numpy twin mixes
float64 into the arithmetic through type promotion (bilinear weights,
kernels.py:190-207) the oracle does the same on purpose.

Semantics chosen where the reference's implementations disagree:
the numpy twins win (they are the only executable reference here):
  * iter_proj stops ALL points when the max step norm over the stop scope drops
    below the threshold (kernels.py:239-241); det is clamped from below
    (kernels.py:227-228).
  * refine_matches re-centres every dilation pass on the INPUT position
    (kernels.py:515), so the result equals the dilation-1 pass.  ``chained=True``
    gives the Metal/CUDA-original behaviour (refine_matches.metal:160-215).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _dot3(a, b):
    """((a0*b0 + a1*b1) + a2*b2) in float32, each op rounded."""
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def normalize_rays(X: np.ndarray) -> np.ndarray:
    """matching.py:121-131: X / sqrt(sum(X*X) + 1e-10), float32."""
    X = np.asarray(X, dtype=F32)
    n2 = _dot3(X, X)
    norm = np.sqrt(n2 + F32(1e-10))
    return X / norm[..., None]


def img_gradient_hwc(img: np.ndarray):
    """image.py:9-34 on a [B,H,W,C] image: central difference / 2, zero border."""
    img = np.asarray(img, dtype=F32)
    gx = np.zeros_like(img)
    gy = np.zeros_like(img)
    gx[:, :, 1:-1, :] = (img[:, :, 2:, :] - img[:, :, :-2, :]) / F32(2.0)
    gy[:, 1:-1, :, :] = (img[:, 2:, :, :] - img[:, :-2, :, :]) / F32(2.0)
    return gx, gy


def prep_for_iter_proj(X11, X21, idx_1_to_2_init=None):
    """matching.py:134-175 -> rays_with_grad [B,H,W,9], pts3d_norm [B,N,3], p_init [B,N,2]."""
    X11 = np.asarray(X11, dtype=F32)
    b, h, w, _ = X11.shape
    rays = normalize_rays(X11)
    gx, gy = img_gradient_hwc(rays)
    rays_with_grad = np.concatenate([rays, gx, gy], axis=-1)
    pts3d_norm = normalize_rays(np.asarray(X21, dtype=F32).reshape(b, -1, 3))
    if idx_1_to_2_init is None:
        idx = np.broadcast_to(np.arange(h * w, dtype=np.int64)[None, :], (b, h * w))
    else:
        idx = np.asarray(idx_1_to_2_init).astype(np.int64)
    p_init = np.stack([idx % w, idx // w], axis=-1).astype(F32)
    return rays_with_grad, pts3d_norm, p_init


def iter_proj(rays_with_grad, pts3d_norm, p_init, max_iter=10, lambda_init=1e-8,
              convergence_thresh=1e-6, stop_scope="global"):
    """kernels.py:151-254.  Returns (p_final [B,N,2] f32, valid [B,N] bool).

    stop_scope: "global" = reference behaviour (max over all B and N);
                "batch"  = per batch item (a batched call == a loop of B=1 calls).
    """
    img = np.asarray(rays_with_grad, dtype=F32)
    tgt = np.asarray(pts3d_norm, dtype=F32)
    b, h, w, _ = img.shape
    n = tgt.shape[1]
    p = np.asarray(p_init).astype(F32).copy()
    lam = F32(lambda_init)
    thr = F32(convergence_thresh)
    xhi = F32(w - 1.001)
    yhi = F32(h - 1.001)
    active = np.ones(b, dtype=bool)            # batch items still iterating
    flat = img.reshape(b, h * w, 9)
    for _ in range(max_iter):
        px = np.minimum(np.maximum(p[..., 0], F32(0)), xhi)
        py = np.minimum(np.maximum(p[..., 1], F32(0)), yhi)
        x0 = np.floor(px).astype(np.int32)
        y0 = np.floor(py).astype(np.int32)
        x1 = np.minimum(x0 + 1, w - 1)
        y1 = np.minimum(y0 + 1, h - 1)
        # float32 - int32 promotes to float64 in the reference (kernels.py:190-191)
        fx = (px.astype(np.float64) - x0.astype(np.float64))[..., None]
        fy = (py.astype(np.float64) - y0.astype(np.float64))[..., None]
        bi = np.arange(b)[:, None]
        v00 = flat[bi, y0 * w + x0].astype(np.float64)
        v01 = flat[bi, y1 * w + x0].astype(np.float64)
        v10 = flat[bi, y0 * w + x1].astype(np.float64)
        v11 = flat[bi, y1 * w + x1].astype(np.float64)
        one = np.float64(1.0)
        s = ((((one - fx) * (one - fy)) * v00 + ((one - fx) * fy) * v01)
             + (fx * (one - fy)) * v10) + (fx * fy) * v11
        s = s.astype(F32)
        ray, gx, gy = s[..., 0:3], s[..., 3:6], s[..., 6:9]
        r = ray - tgt
        a = _dot3(gx, gx) + lam
        bb = _dot3(gx, gy)
        c = _dot3(gy, gx)
        d = _dot3(gy, gy) + lam
        j0 = _dot3(gx, r)
        j1 = _dot3(gy, r)
        det = a * d - bb * c
        det = np.where(det < F32(1e-10), F32(1e-10), det)
        inv_det = F32(1.0) / det
        dx = -(d * j0 - bb * j1) * inv_det
        dy = -((-c) * j0 + a * j1) * inv_det
        pn = p + np.stack([dx, dy], axis=-1)
        p = np.where(active[:, None, None], pn, p)
        dn = np.sqrt(dx * dx + dy * dy)
        with np.errstate(invalid="ignore"):
            if stop_scope == "global":
                if not (np.max(dn) >= thr or np.isnan(np.max(dn))):
                    break
            else:
                mx = np.max(dn, axis=1)
                active &= ~(mx < thr)
                if not active.any():
                    break
    p_final = np.stack([
        np.minimum(np.maximum(p[..., 0], F32(0)), F32(w - 1)),
        np.minimum(np.maximum(p[..., 1], F32(0)), F32(h - 1)),
    ], axis=-1)
    valid = (p[..., 0] >= 0) & (p[..., 0] < w) & (p[..., 1] >= 0) & (p[..., 1] < h)
    return p_final, valid


def _refine_pass(D11, D21, cx, cy, radius, dil):
    """One window search.  Score = sequential float32 sum_d (q[d]*r[d]) (mul then
    add, each rounded); strict '>' in (dy outer, dx inner) raster order."""
    b, h, w, dd = D11.shape
    n = D21.shape[1]
    best = np.full((b, n), -np.inf, dtype=F32)
    bx, by = cx.copy(), cy.copy()
    flat = D11.reshape(b, h * w, dd)
    bi = np.arange(b)[:, None]
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            ny = cy + dy * dil
            nx = cx + dx * dil
            inb = (nx >= 0) & (nx < w) & (ny >= 0) & (ny < h)
            lin = np.clip(ny, 0, h - 1) * w + np.clip(nx, 0, w - 1)
            ref = flat[bi, lin]                       # [B,N,D]
            score = np.zeros((b, n), dtype=F32)
            for k in range(dd):
                score = score + D21[..., k] * ref[..., k]
            with np.errstate(invalid="ignore"):
                upd = inb & (score > best)
            best = np.where(upd, score, best)
            bx = np.where(upd, nx, bx)
            by = np.where(upd, ny, by)
    return bx, by


def refine_matches(D11, D21, p1, radius=3, dilation_max=0, chained=False):
    """kernels.py:496-537.  p1 [B,N,2] (int, or float -> truncated).  Returns int32 [B,N,2]."""
    D11 = np.asarray(D11, dtype=F32)
    D21 = np.asarray(D21, dtype=F32)
    p1i = np.trunc(np.asarray(p1)).astype(np.int32)
    cx0, cy0 = p1i[..., 0].astype(np.int64), p1i[..., 1].astype(np.int64)
    rx, ry = cx0.copy(), cy0.copy()
    for dil in range(max(1, int(dilation_max)), 0, -1):
        if chained:
            rx, ry = _refine_pass(D11, D21, rx, ry, radius, dil)
        elif dil == 1:
            # every pass re-centres on the input, the last (dil=1) overwrites the rest
            rx, ry = _refine_pass(D11, D21, cx0, cy0, radius, dil)
    return np.stack([rx, ry], axis=-1).astype(np.int32)


def match_epilogue(X11, X21, p_int, valid_proj, dist_thresh=0.1):
    """matching.py:436-461: gather X11 at clipped p, 3-D distance test, linear index."""
    X11 = np.asarray(X11, dtype=F32)
    b, h, w, _ = X11.shape
    X21f = np.asarray(X21, dtype=F32).reshape(b, h * w, 3)
    p_int = np.asarray(p_int).astype(np.int64)
    y = np.clip(p_int[..., 1], 0, h - 1)
    x = np.clip(p_int[..., 0], 0, w - 1)
    lin = y * w + x
    bi = np.arange(b)[:, None]
    d = X11.reshape(b, h * w, 3)[bi, lin] - X21f
    dist = np.sqrt(_dot3(d, d))
    valid = np.asarray(valid_proj, dtype=bool) & (dist < F32(dist_thresh))
    idx = p_int[..., 0] + w * p_int[..., 1]
    return idx.astype(np.int64), valid[..., None]


def match_simple(X11, X21, idx_1_to_2_init=None, dist_thresh=0.1):
    """matching.py:41-90."""
    X11 = np.asarray(X11, dtype=F32)
    b, h, w, _ = X11.shape
    n = h * w
    if idx_1_to_2_init is None:
        idx = np.broadcast_to(np.arange(n, dtype=np.int64)[None, :], (b, n)).copy()
    else:
        idx = np.asarray(idx_1_to_2_init).astype(np.int64)
    bi = np.arange(b)[:, None]
    d = X11.reshape(b, n, 3)[bi, idx] - np.asarray(X21, dtype=F32).reshape(b, n, 3)
    dist = np.sqrt(_dot3(d, d))
    return idx, (dist < F32(dist_thresh))[..., None]


def match_iterative_proj(X11, X21, D11, D21, idx_1_to_2_init=None, *, max_iter=10,
                         lambda_init=1e-8, convergence_thresh=1e-6, dist_thresh=0.1,
                         radius=3, dilation_max=2, chained=False, stop_scope="batch"):
    """matching.py:339-461 end to end (prep -> iter_proj -> refine -> epilogue)."""
    b, h, w, _ = np.asarray(X11).shape
    rays, tgt, p0 = prep_for_iter_proj(X11, X21, idx_1_to_2_init)
    p, valid_proj = iter_proj(rays, tgt, p0, max_iter, lambda_init, convergence_thresh, stop_scope)
    p_int = p.astype(np.int32)
    if radius > 0:
        D21f = np.asarray(D21, dtype=F32).reshape(b, h * w, -1)
        p_int = refine_matches(D11, D21f, p_int, radius, dilation_max, chained)
    return match_epilogue(X11, X21, p_int, valid_proj, dist_thresh)


# ---------------------------------------------------------------------------------------------------
# Fast reciprocal NN (own semantics - the reference tree has no implementation, SURVEY 8a row K8;
# algorithm: MASt3R, Leroy et al. 2024, section 3.3).  Scores in float64; the device computes an fp32
# FMA chain, so tests compare indices exactly where the float64 margin between the best and the second
# best candidate exceeds the fp32 rounding bound, and by score otherwise.
def nn_search(Q, DB, chunk=1024):
    """Q [S,D], DB [N,D] -> (idx [S] lowest argmax of Q.DB^T in float64, best [S], second best [S])."""
    Q64, DB64 = np.asarray(Q, np.float64), np.asarray(DB, np.float64)
    idx = np.empty(len(Q64), np.int64); best = np.empty(len(Q64)); second = np.empty(len(Q64))
    for lo in range(0, len(Q64), chunk):
        sc = Q64[lo:lo + chunk] @ DB64.T
        i = np.argmax(sc, axis=1)                       # first maximum = lowest index
        idx[lo:lo + chunk] = i
        best[lo:lo + chunk] = sc[np.arange(len(i)), i]
        sc[np.arange(len(i)), i] = -np.inf
        second[lo:lo + chunk] = sc.max(axis=1)
    return idx, best, second


def fast_reciprocal_nn(D1, D2, subsample=8, max_iter=10):
    h1, w1, d = D1.shape
    f1, f2 = D1.reshape(-1, d), D2.reshape(-1, d)
    ys, xs = np.arange(subsample // 2, h1, subsample), np.arange(subsample // 2, w1, subsample)
    xy1 = (ys[:, None] * w1 + xs[None, :]).reshape(-1)
    pairs = []
    for _ in range(max_iter):
        if len(xy1) == 0:
            break
        xy2 = nn_search(f1[xy1], f2)[0]
        back = nn_search(f2[xy2], f1)[0]
        conv = back == xy1
        pairs.append(np.stack([xy1[conv], xy2[conv]], 1))
        xy1 = np.unique(back[~conv])
    if not pairs:
        return np.empty(0, np.int64), np.empty(0, np.int64)
    p = np.unique(np.concatenate(pairs), axis=0)
    return p[:, 0], p[:, 1]
