"""Generate tests/golden/*.npz from the reference's importable numpy twins.

Run ONLY in the build container (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
The fixtures hold inputs + the reference's outputs (data, not source).  The GPU
box never sees /root/reference; tests read only the committed .npz files.

Reference entry points exercised (paths under /root/reference/src/mlx_mast3r_slam):
  backends/mpsgraph/kernels.py      _iter_proj_numpy :151, _refine_matches_numpy :496
  backends/mpsgraph/gauss_newton.py gauss_newton_rays :23
  backends/mpsgraph/gauss_newton_points.py gauss_newton_points :17
  backends/mpsgraph/gauss_newton_calib.py gauss_newton_calib :17
  backends/mpsgraph/sim3_ops.py     quat_multiply, quat_rotate, sim3_relative, exp_so3,
                                    exp_sim3, retract_sim3, huber_weight
  backends/mpsgraph/linalg.py       cholesky_solve :17
  /root/reference/benchmark_all_kernels.py create_gn_test_data :16 (input recipe)
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(ROOT, "mast3r-slam_amd"))
sys.path.insert(0, ROOT)

from mlx_mast3r_slam.backends.mpsgraph import kernels as rk            # noqa: E402
from mlx_mast3r_slam.backends.mpsgraph import gauss_newton as rgn     # noqa: E402
from mlx_mast3r_slam.backends.mpsgraph import gauss_newton_points as rgp  # noqa: E402
from mlx_mast3r_slam.backends.mpsgraph import gauss_newton_calib as rgc   # noqa: E402
from mlx_mast3r_slam.backends.mpsgraph import sim3_ops as rs          # noqa: E402
from mlx_mast3r_slam.backends.mpsgraph import linalg as rl            # noqa: E402

from mast3r_slam import synthetic                                      # noqa: E402
from oracle import matching as om                                      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _bench_module():
    spec = importlib.util.spec_from_file_location("ref_bench", os.path.join(REF, "benchmark_all_kernels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    os.makedirs(OUT, exist_ok=True)

    # ---- 1. iter_proj on a smooth two-view scene (B=1 and B=2) -------------
    for tag, b, h, w in (("b1", 1, 48, 64), ("b2", 2, 40, 56)):
        sc = synthetic.geometric_pair(h, w, seed=7, batch=b)
        rays, tgt, p0 = om.prep_for_iter_proj(sc["X11"], sc["X21"], None)
        p_ref, v_ref = rk._iter_proj_numpy(rays, tgt, p0, 10, 1e-8, 1e-6)
        np.savez_compressed(os.path.join(OUT, f"iter_proj_{tag}.npz"),
                            rays_with_grad=rays, pts3d_norm=tgt, p_init=p0,
                            p_ref=p_ref.astype(np.float32), valid_ref=v_ref,
                            max_iter=10, lambda_init=1e-8, convergence_thresh=1e-6)
    # early-stop case: a sub-pixel warp that vanishes at the image border, so every point
    # converges and the global max step norm really drops below the (large) threshold
    h, w = 24, 32
    vv, uu = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    u1 = uu + 0.4 * np.sin(np.pi * uu / (w - 1)) * np.sin(np.pi * vv / (h - 1))
    v1 = vv - 0.3 * np.sin(np.pi * uu / (w - 1)) * np.sin(np.pi * vv / (h - 1))
    X11 = synthetic._surface(uu, vv, h, w)[None].astype(np.float32)
    X21 = synthetic._surface(u1, v1, h, w)[None].astype(np.float32)
    rays, tgt, p0 = om.prep_for_iter_proj(X11, X21, None)
    p_ref, v_ref = rk._iter_proj_numpy(rays, tgt, p0, 10, 1e-8, 0.05)
    np.savez_compressed(os.path.join(OUT, "iter_proj_earlystop.npz"),
                        rays_with_grad=rays, pts3d_norm=tgt, p_init=p0,
                        p_ref=p_ref.astype(np.float32), valid_ref=v_ref,
                        max_iter=10, lambda_init=1e-8, convergence_thresh=0.05)

    # ---- 2. refine_matches ---------------------------------------------------
    sc = synthetic.geometric_pair(48, 64, seed=11, batch=1)
    rng = np.random.default_rng(5)
    n = 48 * 64
    p1 = np.rint(sc["uv_true"] + rng.normal(0, 1.5, size=(1, n, 2))).astype(np.int32)
    p1[0, :40, 0] = rng.integers(-3, 3, 40)           # some out-of-bounds / border centres
    p1[0, 40:80, 1] = rng.integers(46, 52, 40)
    D21 = sc["D21"].reshape(1, n, -1)
    for dmax in (0, 2):
        ref = rk._refine_matches_numpy(sc["D11"], D21, p1, 3, dmax)
        np.savez_compressed(os.path.join(OUT, f"refine_matches_d{dmax}.npz"),
                            D11=sc["D11"], D21=D21, p1=p1, p_ref=ref.astype(np.int32),
                            radius=3, dilation_max=dmax)

    # ---- 2b. edge cases (round 4): inputs the smooth scenes above never produce ----------------------------------
    # iter_proj: random (non-smooth) ray map -> steps that leave the image, the determinant clamp, points that start outside
    # the image or on its border, a subset of points (N != H * W), one and three iterations, a large damping
    rng = np.random.default_rng(21)
    h, w, n = 20, 28, 300
    rays = rng.normal(size=(2, h, w, 9)).astype(np.float32)
    rays[..., :3] /= np.linalg.norm(rays[..., :3], axis=-1, keepdims=True)
    tgt = rng.normal(size=(2, n, 3)).astype(np.float32)
    tgt /= np.linalg.norm(tgt, axis=-1, keepdims=True)
    p0 = np.stack([rng.uniform(-3.0, w + 2.0, size=(2, n)), rng.uniform(-3.0, h + 2.0, size=(2, n))], -1).astype(np.float32)
    p0[0, :10] = np.array([0.0, 0.0], np.float32); p0[0, 10:20] = np.array([w - 1.0, h - 1.0], np.float32)
    p0[1, :10, 0] = w - 1.001
    for tag, iters, lam in (("edge_it1", 1, 1e-8), ("edge_it3", 3, 1e-8), ("edge_lam", 10, 1e-2)):
        p_ref, v_ref = rk._iter_proj_numpy(rays, tgt, p0, iters, lam, 1e-6)
        np.savez_compressed(os.path.join(OUT, f"iter_proj_{tag}.npz"), rays_with_grad=rays, pts3d_norm=tgt, p_init=p0,
                            p_ref=p_ref.astype(np.float32), valid_ref=v_ref, max_iter=iters, lambda_init=lam,
                            convergence_thresh=1e-6)
    # the reference benchmark's OWN input recipe for iter_proj (benchmark_all_kernels.py:56-77, its first configuration:
    # b = 1, 64 x 64, 1000 points; un-normalised random rays, random unit targets, uniform starts) under its seed
    np.random.seed(42)
    b_, h_, w_, n_ = 1, 64, 64, 1000
    rays_b = np.random.randn(b_, h_, w_, 9).astype(np.float32)
    pts_b = np.random.randn(b_, n_, 3).astype(np.float32)
    pts_b = pts_b / np.linalg.norm(pts_b, axis=-1, keepdims=True)
    p_b = np.stack([np.random.rand(b_, n_) * (w_ - 1), np.random.rand(b_, n_) * (h_ - 1)], axis=-1).astype(np.float32)
    p_ref, v_ref = rk._iter_proj_numpy(rays_b, pts_b, p_b.copy(), 10, 1e-8, 1e-6)
    np.savez_compressed(os.path.join(OUT, "iter_proj_refbench.npz"), rays_with_grad=rays_b, pts3d_norm=pts_b, p_init=p_b,
                        p_ref=p_ref.astype(np.float32), valid_ref=v_ref, max_iter=10, lambda_init=1e-8, convergence_thresh=1e-6)
    # refine_matches: other descriptor lengths and radii, random descriptors, centres on and beyond every border, ties
    for tag, d, radius in (("r2_d16", 16, 2), ("r4_d24", 24, 4), ("r1_d64", 64, 1), ("r3_d5", 5, 3)):
        rng = np.random.default_rng(100 + d)
        b, h, w, n = 2, 22, 31, 500
        D11 = rng.normal(size=(b, h, w, d)).astype(np.float32)
        D11[1, 5:9, 7:12] = D11[1, 5, 7]                     # a constant patch: ties inside the window
        D21 = rng.normal(size=(b, n, d)).astype(np.float32)
        p1 = np.stack([rng.integers(-5, w + 5, size=(b, n)), rng.integers(-5, h + 5, size=(b, n))], -1).astype(np.int32)
        p1[1, :40] = np.stack([rng.integers(6, 12, 40), rng.integers(4, 9, 40)], -1)
        ref = rk._refine_matches_numpy(D11, D21, p1, radius, 2)
        np.savez_compressed(os.path.join(OUT, f"refine_matches_{tag}.npz"), D11=D11, D21=D21, p1=p1, p_ref=ref.astype(np.int32),
                            radius=radius, dilation_max=2)

    # ---- 3. gauss_newton_rays on the reference's own generator ----------------
    rb = _bench_module()
    np.random.seed(42)
    Twc, Xs, Cs, ii, jj, idx, valid, Q = rb.create_gn_test_data(5, 200, 8)
    for it in (1, 3):
        out = rgn.gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx, valid, Q, max_iter=it, pin=1)
        np.savez_compressed(os.path.join(OUT, f"gn_rays_it{it}.npz"), Twc=Twc, Xs=Xs, Cs=Cs, ii=ii, jj=jj,
                            idx=idx, valid=valid, Q=Q, Twc_ref=out, max_iter=it, pin=1)
    for it in (1, 3):
        out = rgp.gauss_newton_points(Twc, Xs, Cs, ii, jj, idx, valid, Q, max_iter=it, pin=1)
        np.savez_compressed(os.path.join(OUT, f"gn_points_it{it}.npz"), Twc=Twc, Xs=Xs, Cs=Cs, ii=ii, jj=jj,
                            idx=idx, valid=valid, Q=Q, Twc_ref=out, max_iter=it, pin=1)
    # calibrated variant: inputs as in benchmark_all_kernels.py:215-227 (positive depth, K, 640x480)
    Kmat = np.array([[500, 0, 320], [0, 500, 240], [0, 0, 1]], dtype=np.float32)
    Xs_pos = np.abs(Xs) + 0.1
    for it in (1, 3):
        out = rgc.gauss_newton_calib(Twc.copy(), Xs_pos, Cs, Kmat, ii, jj, idx, valid, Q, (640, 480), max_iter=it, pin=1)
        np.savez_compressed(os.path.join(OUT, f"gn_calib_it{it}.npz"), Twc=Twc, Xs=Xs_pos, Cs=Cs, K=Kmat, ii=ii, jj=jj,
                            idx=idx, valid=valid, Q=Q, Twc_ref=out, max_iter=it, pin=1, img_size=np.array([640, 480]))
    # calibrated, well-conditioned: chain graph seen by perturbed poses, all points in front of the cameras
    Tc, Xc, Cc, iic, jjc, idxc, validc, Qc = synthetic.gn_graph(6, 150, 0, seed=13, chain=True, pose_noise=0.01)
    Tc[:, :3] *= 0.2                                    # keep the cloud (z ~ 4) inside every 640x480 view
    Tc[:, 3:7] = np.array([0, 0, 0, 1]) + 0.02 * np.random.default_rng(3).normal(size=(6, 4))
    Tc[:, 3:7] /= np.linalg.norm(Tc[:, 3:7], axis=1, keepdims=True)
    _, Xc, _, _, _, _, _, _ = synthetic.gn_graph(6, 150, 0, seed=13, chain=True, pose_noise=0.0, poses=Tc)
    Tn = Tc.copy(); Tn[1:, :3] += 0.01 * np.random.default_rng(4).normal(size=(5, 3)).astype(np.float32)
    out = rgc.gauss_newton_calib(Tn.copy(), Xc, Cc, Kmat, iic, jjc, idxc, validc, Qc, (640, 480), max_iter=8, pin=1)
    np.savez_compressed(os.path.join(OUT, "gn_calib_chain.npz"), Twc=Tn, Xs=Xc, Cs=Cc, K=Kmat, ii=iic, jj=jjc, idx=idxc,
                        valid=validc, Q=Qc, Twc_ref=out, max_iter=8, pin=1, img_size=np.array([640, 480]))
    # a solvable chain graph (converges), own generator
    Twc, Xs, Cs, ii, jj, idx, valid, Q = synthetic.gn_graph(6, 150, 0, seed=9, chain=True, pose_noise=0.02)
    out = rgn.gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx, valid, Q, max_iter=10, pin=1)
    np.savez_compressed(os.path.join(OUT, "gn_rays_chain.npz"), Twc=Twc, Xs=Xs, Cs=Cs, ii=ii, jj=jj,
                        idx=idx, valid=valid, Q=Q, Twc_ref=out, max_iter=10, pin=1)

    # the same kind of graph with every optional argument off its default (round 4): two pinned poses, thresholds that cut
    # points (confidence and match quality), another sigma, a convergence threshold that stops the loop early, and a graph
    # with a repeated and a reversed edge
    Twc, Xs, Cs, ii, jj, idx, valid, Q = synthetic.gn_graph(7, 180, 2, seed=17, chain=True, pose_noise=0.015)
    ii = np.concatenate([ii, ii[:1], jj[1:2]]).astype(ii.dtype); jj2 = np.concatenate([jj, jj[:1], ii[1:2]]).astype(jj.dtype)
    idx = np.concatenate([idx, idx[:1], idx[1:2]]); valid = np.concatenate([valid, valid[:1], valid[1:2]]); Q = np.concatenate([Q, Q[:1], Q[1:2]])
    kw = dict(sigma_ray=0.01, sigma_dist=5.0, C_thresh=0.6, Q_thresh=2.0, max_iter=6, delta_thresh=2e-3, pin=2)
    out = rgn.gauss_newton_rays(Twc, Xs, Cs, ii, jj2, idx, valid, Q, **kw)
    np.savez_compressed(os.path.join(OUT, "gn_rays_params.npz"), Twc=Twc, Xs=Xs, Cs=Cs, ii=ii, jj=jj2, idx=idx, valid=valid, Q=Q,
                        Twc_ref=out, **kw)
    out = rgp.gauss_newton_points(Twc, Xs, Cs, ii, jj2, idx, valid, Q, sigma_point=0.02, C_thresh=0.6, Q_thresh=2.0, max_iter=6,
                                  delta_thresh=2e-3, pin=2)
    np.savez_compressed(os.path.join(OUT, "gn_points_params.npz"), Twc=Twc, Xs=Xs, Cs=Cs, ii=ii, jj=jj2, idx=idx, valid=valid, Q=Q,
                        Twc_ref=out, sigma_point=0.02, C_thresh=0.6, Q_thresh=2.0, max_iter=6, delta_thresh=2e-3, pin=2)

    # calibrated variant with its optional arguments off their defaults: a pixel border and a depth floor that reject
    # projections, other sigmas, thresholds, two pinned poses (the well-conditioned chain graph of gn_calib_chain)
    kwc = dict(pixel_border=40, z_eps=0.05, sigma_pixel=2.0, sigma_depth=0.3, C_thresh=0.6, Q_thresh=2.0, max_iter=5,
               delta_thresh=1e-3, pin=2)
    out = rgc.gauss_newton_calib(Tn.copy(), Xc, Cc, Kmat, iic, jjc, idxc, validc, Qc, (640, 480), **kwc)
    np.savez_compressed(os.path.join(OUT, "gn_calib_params.npz"), Twc=Tn, Xs=Xc, Cs=Cc, K=Kmat, ii=iic, jj=jjc, idx=idxc,
                        valid=validc, Q=Qc, Twc_ref=out, img_size=np.array([640, 480]), **kwc)

    # ---- 4. sim3_ops known answers -------------------------------------------------
    rng = np.random.default_rng(1)
    q1 = rng.normal(size=(16, 4)); q1 /= np.linalg.norm(q1, axis=-1, keepdims=True)
    q2 = rng.normal(size=(16, 4)); q2 /= np.linalg.norm(q2, axis=-1, keepdims=True)
    v = rng.normal(size=(16, 3))
    t1, t2 = rng.normal(size=(16, 3)), rng.normal(size=(16, 3))
    s1, s2 = rng.uniform(0.5, 2, 16), rng.uniform(0.5, 2, 16)
    xi = rng.normal(size=(16, 7)) * 0.3
    xi[0] = 0.0                      # small theta, small sigma
    xi[1, 3:6] = 1e-5                # small theta, large sigma
    xi[2, 6] = 1e-8                  # large theta, small sigma
    xi[3, 3:6] *= 8.0                # big rotation
    tij, qij, sij = rs.sim3_relative(t1, q1, s1, t2, q2, s2)
    et, eq, es = rs.exp_sim3(xi)
    rt, rq, rsc = rs.retract_sim3(xi, t1, q1, s1)
    r = rng.normal(size=64) * 3
    np.savez_compressed(os.path.join(OUT, "sim3_ops.npz"), q1=q1, q2=q2, v=v, t1=t1, t2=t2, s1=s1, s2=s2, xi=xi,
                        qmul=rs.quat_multiply(q1, q2), qrot=rs.quat_rotate(q1, v),
                        rel_t=tij, rel_q=qij, rel_s=sij, exp_so3=rs.exp_so3(xi[:, 3:6]),
                        exp_t=et, exp_q=eq, exp_s=es, retr_t=rt, retr_q=rq, retr_s=rsc,
                        hub_r=r, hub_w=rs.huber_weight(r))

    # ---- 5. linalg.cholesky_solve 7x7 -------------------------------------------------
    A = rng.normal(size=(40, 7))
    H = A.T @ A
    g = rng.normal(size=7)
    np.savez_compressed(os.path.join(OUT, "cholesky_solve.npz"), H=H, g=g, x=rl.cholesky_solve(H, g, 1e-6),
                        H32=H.astype(np.float32), g32=g.astype(np.float32),
                        x32=rl.cholesky_solve(H.astype(np.float32), g.astype(np.float32), 1e-6))
    # a backend-sized system (30 free keyframes = 210 unknowns: several 64-wide block columns of the device Cholesky) with
    # the conditioning of a pose-graph normal matrix (eigenvalues over 6 decades)
    rng = np.random.default_rng(2)
    n = 210
    U, _ = np.linalg.qr(rng.normal(size=(n, n)))
    H = (U * np.logspace(0, 6, n)) @ U.T
    H = 0.5 * (H + H.T)
    g = rng.normal(size=n) * 100.0
    np.savez_compressed(os.path.join(OUT, "cholesky_solve_n210.npz"), H=H, g=g, x=rl.cholesky_solve(H, g, 1e-6))
    # ---- 5b. the sim3_ops helpers the first fixture left out: quat_inv, sim3_act, huber_weight with another k -------------
    rng = np.random.default_rng(31)
    q = rng.normal(size=(16, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.normal(size=(16, 3)); sc_ = np.exp(rng.normal(size=(16, 1)) * 0.2)
    X = rng.normal(size=(16, 50, 3))
    r = rng.normal(size=128) * 4
    np.savez_compressed(os.path.join(OUT, "sim3_ops_more.npz"), q=q, t=t, s=sc_, X=X, r=r,
                        qinv=rs.quat_inv(q), act=rs.sim3_act(t[:, None, :], q[:, None, :], sc_[:, None, :], X),
                        hub_k2=rs.huber_weight(r, 2.0), hub_k05=rs.huber_weight(r, 0.5))

    # ---- 6. the reference's DEFAULT_CONFIG (config.py:53-...) as data: the constants the hot path reads must be the
    # reference's (tests/test_abi_and_host.py compares mast3r_slam.config.DEFAULT_CONFIG key by key)
    import json
    from mlx_mast3r_slam import config as rcfg                                # imports pathlib / yaml only
    with open(os.path.join(OUT, "reference_default_config.json"), "w") as f:
        json.dump(rcfg.DEFAULT_CONFIG, f, indent=1, sort_keys=True)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f:32s} {os.path.getsize(os.path.join(OUT, f)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
