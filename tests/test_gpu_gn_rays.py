"""GPU parity: backend Gauss-Newton ("rays") blocks and solve vs the float64 oracle and the
golden vectors frozen from the reference's numpy twin."""
import os

import numpy as np
import pytest
import torch

from mast3r_slam import kernels, synthetic
from oracle import gn_rays as og

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_edge_blocks_match_oracle(golden_dir):
    z = _load(golden_dir, "gn_rays_it1.npz")
    blocks = kernels.gn_rays_blocks(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"])
    t = z["Twc"][:, :3].astype(np.float64); q = z["Twc"][:, 3:7].astype(np.float64); s = z["Twc"][:, 7].astype(np.float64)
    iu = np.triu_indices(7)
    for e in range(len(z["ii"])):
        Hjj, gj, n = og.edge_blocks(t, q, s, z["Xs"], z["Cs"], int(z["ii"][e]), int(z["jj"][e]), z["idx"][e],
                                    z["valid"][e], z["Q"][e])
        assert blocks[e, 35] == n                                          # integer count: exact
        assert np.abs(blocks[e, :28] - Hjj[iu]).max() <= 3e-5 * np.abs(Hjj).max()   # float32 per-point terms
        assert np.abs(blocks[e, 28:35] - gj).max() <= 3e-5 * np.abs(gj).max()


@pytest.mark.parametrize("tag,tol", [("it1", 2e-5), ("it3", 5e-4), ("chain", 5e-5)])
def test_gauss_newton_rays_golden(golden_dir, tag, tol):
    """it3 is the reference benchmark's random (non-convergent, |dx|~0.5/iter) problem: float32 per-point
    arithmetic differences are amplified over iterations, hence the looser bound there."""
    z = _load(golden_dir, f"gn_rays_{tag}.npz")
    out, info = kernels.gauss_newton_rays(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                          max_iter=int(z["max_iter"]), pin=int(z["pin"]), return_info=True)
    assert isinstance(out, np.ndarray) and out.dtype == np.float32          # numpy in -> numpy out
    assert not info["failed"]
    assert np.abs(out - z["Twc_ref"]).max() <= tol, np.abs(out - z["Twc_ref"]).max()
    pinned = int(np.unique(np.concatenate([z["ii"], z["jj"]]))[0])
    assert np.array_equal(out[pinned], z["Twc"][pinned])


def test_gauss_newton_goldens_with_every_argument_off_its_default(golden_dir):
    """Outputs of the REFERENCE's twins with sigma / C_thresh / Q_thresh / max_iter / delta_thresh / pin all off their
    defaults (two pinned poses, thresholds that cut points, an early stop, a repeated and a reversed edge)."""
    z = _load(golden_dir, "gn_rays_params.npz")
    kw = {k: (int(z[k]) if k in ("max_iter", "pin") else float(z[k])) for k in
          ("sigma_ray", "sigma_dist", "C_thresh", "Q_thresh", "max_iter", "delta_thresh", "pin")}
    out, info = kernels.gauss_newton_rays(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                          return_info=True, **kw)
    assert not info["failed"] and np.abs(out - z["Twc_ref"]).max() <= 5e-5, np.abs(out - z["Twc_ref"]).max()
    assert np.array_equal(out[:2], z["Twc"][:2])
    z = _load(golden_dir, "gn_points_params.npz")
    kw = {k: (int(z[k]) if k in ("max_iter", "pin") else float(z[k])) for k in
          ("sigma_point", "C_thresh", "Q_thresh", "max_iter", "delta_thresh", "pin")}
    out, info = kernels.gauss_newton_points(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                            return_info=True, **kw)
    assert not info["failed"] and np.abs(out - z["Twc_ref"]).max() <= 5e-5, np.abs(out - z["Twc_ref"]).max()
    assert np.array_equal(out[:2], z["Twc"][:2])


def test_gauss_newton_calib_golden_with_every_argument_off_its_default(golden_dir):
    """Reference twin of the calibrated variant with a pixel border and a depth floor that reject projections, other sigmas
    and thresholds, two pinned poses, an early stop."""
    z = _load(golden_dir, "gn_calib_params.npz")
    kw = {k: (int(z[k]) if k in ("max_iter", "pin", "pixel_border") else float(z[k])) for k in
          ("pixel_border", "z_eps", "sigma_pixel", "sigma_depth", "C_thresh", "Q_thresh", "max_iter", "delta_thresh", "pin")}
    out, info = kernels.gauss_newton_calib(z["Twc"], z["Xs"], z["Cs"], z["K"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                           tuple(int(v) for v in z["img_size"]), return_info=True, **kw)
    assert not info["failed"] and np.abs(out - z["Twc_ref"]).max() <= 5e-5, np.abs(out - z["Twc_ref"]).max()
    assert np.array_equal(out[:2], z["Twc"][:2])


def test_cholesky_solve_golden_backend_sized_system(dev, golden_dir):
    """kernels.cholesky_solve (the blocked float64 device Cholesky, 4 block columns here) against the output of the
    REFERENCE's linalg.cholesky_solve on a 210-unknown system with eigenvalues over 6 decades."""
    z = _load(golden_dir, "cholesky_solve_n210.npz")
    x = kernels.cholesky_solve(z["H"], z["g"], 1e-6)
    assert isinstance(x, np.ndarray) and x.dtype == np.float64
    assert np.abs(x - z["x"]).max() <= 1e-8 * np.abs(z["x"]).max(), np.abs(x - z["x"]).max() / np.abs(z["x"]).max()


def test_degenerate_graphs_return_input(dev):
    Twc, Xs, Cs, ii, jj, idx, valid, Q = synthetic.gn_graph(4, 64, 3, seed=1)
    out = kernels.gauss_newton_rays(Twc, Xs, Cs, ii[:0], jj[:0], idx[:0], valid[:0], Q[:0])
    assert np.array_equal(out, Twc)
    out = kernels.gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx, valid, Q, pin=4)
    assert np.array_equal(out, Twc)
    # no valid point on any edge -> H = 1e-6 I, dx = 0 -> stops before any update
    out, info = kernels.gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx, np.zeros_like(valid), Q, return_info=True)
    assert np.array_equal(out, Twc) and info["iters"] == 0 and info["stopped"]


def test_device_tensors_and_large_graph_path_agree(dev):
    # 70 keyframes -> dim 483 > single-workgroup limit: exercises the assemble/retract + library-solve path
    Twc, Xs, Cs, ii, jj, idx, valid, Q = synthetic.gn_graph(70, 256, 0, seed=3, chain=True, pose_noise=0.01)
    args = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (Twc, Xs, Cs, ii, jj, idx, valid, Q)]
    out, info = kernels.gauss_newton_rays(*args, max_iter=4, return_info=True)
    assert isinstance(out, torch.Tensor) and out.is_cuda and not info["failed"]
    ref = og.gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx, valid, Q, max_iter=4)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-4
    # and a graph small enough for the on-device Cholesky gives the same answer as the oracle
    Twc, Xs, Cs, ii, jj, idx, valid, Q = synthetic.gn_graph(20, 512, 0, seed=4, chain=True, pose_noise=0.01)
    out = kernels.gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx, valid, Q, max_iter=5)
    ref = og.gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx, valid, Q, max_iter=5)
    assert np.abs(out - ref).max() < 2e-4


@pytest.mark.parametrize("tag,tol", [("it1", 2e-5), ("it3", 5e-4)])
def test_gauss_newton_points_golden(golden_dir, tag, tol):
    """kernels.gauss_newton_points vs the reference's numpy twin (gauss_newton_points.py)."""
    z = _load(golden_dir, f"gn_points_{tag}.npz")
    out, info = kernels.gauss_newton_points(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                            max_iter=int(z["max_iter"]), pin=int(z["pin"]), return_info=True)
    assert not info["failed"]
    assert np.abs(out - z["Twc_ref"]).max() <= tol, np.abs(out - z["Twc_ref"]).max()
    blocks_r = kernels.gn_rays_blocks(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"], 0.01)
    blocks_p = kernels.gn_rays_blocks(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"], 0.01,
                                      point_mode=True)
    assert not np.allclose(blocks_r[:, :28], blocks_p[:, :28])            # the extra weight is really applied
    assert np.array_equal(blocks_r[:, 35], blocks_p[:, 35])


def test_gauss_newton_calib_golden(golden_dir):
    """kernels.gauss_newton_calib vs the reference's numpy twin (gauss_newton_calib.py).
    (a) a well-conditioned chain graph (cond(H) ~ 3e4, converges): tight comparison;
    (b) the reference benchmark's own random recipe: one keyframe there is constrained by 2 points only
        (cond(H) ~ 4e11, |dx| ~ 80-170 - the reference's doc reports NaN/divergence for this kernel), so
        only the well-observed keyframes are compared and the rest must stay finite."""
    zc = _load(golden_dir, "gn_calib_chain.npz")
    size = tuple(int(v) for v in zc["img_size"])
    out, info = kernels.gauss_newton_calib(zc["Twc"], zc["Xs"], zc["Cs"], zc["K"], zc["ii"], zc["jj"], zc["idx"],
                                           zc["valid"], zc["Q"], size, max_iter=int(zc["max_iter"]), pin=1, return_info=True)
    assert not info["failed"]
    assert np.abs(out - zc["Twc_ref"]).max() <= 2e-5, np.abs(out - zc["Twc_ref"]).max()
    assert np.abs(zc["Twc_ref"] - zc["Twc"]).max() > 1e-3
    z = _load(golden_dir, "gn_calib_it1.npz")
    args = (z["Twc"], z["Xs"], z["Cs"], z["K"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"], size)
    blocks = kernels.gn_rays_blocks(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"], 1.0)
    out, info = kernels.gauss_newton_calib(*args, max_iter=1, pin=1, return_info=True)
    assert not info["failed"] and np.isfinite(out).all()
    assert np.abs(out[4] - z["Twc_ref"][4]).max() <= 2e-5                  # keyframe 4: 54 valid observations
    assert np.array_equal(out[[0, 1, 3]], z["Twc"][[0, 1, 3]])              # pinned / unobserved keyframes untouched
    # per-edge blocks of the calibrated residual against the float64 oracle
    t = z["Twc"][:, :3].astype(np.float64); q = z["Twc"][:, 3:7].astype(np.float64); s = z["Twc"][:, 7].astype(np.float64)
    calib = dict(fx=500.0, fy=500.0, cx=320.0, cy=240.0, width=640, height=480, border=0, z_eps=0.0, sigma_pixel=1.0,
                 sigma_depth=0.1)
    import ctypes  # noqa: F401
    from mast3r_slam import _ffi
    tt = kernels._prep_gn(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"])
    e, p, k = tt["E"], tt["P"], tt["K"]
    bl = torch.empty((e, 36), dtype=torch.float64, device=tt["Twc"].device)
    ws = torch.empty(e * _ffi.lib().m3_gn_rays_chunks(p) * 36, dtype=torch.float64, device=tt["Twc"].device)
    _ffi.call("m3_gn_rays_blocks", _ffi.ptr(tt["Twc"]), _ffi.ptr(tt["Xs"]), _ffi.ptr(tt["Cs"]), _ffi.ptr(tt["ii"]),
              _ffi.ptr(tt["jj"]), _ffi.ptr(tt["idx"]), _ffi.ptr(tt["valid"]), _ffi.ptr(tt["Q"]), _ffi.ptr(bl), _ffi.ptr(ws),
              k, p, e, 1.0, 0.0, 1.5, 2,
              kernels._calib_ptr((500, 500, 320, 240, 640, 480, 0, 0.0, 1.0, 0.1)), _ffi.stream_ptr())
    bl = bl.cpu().numpy()
    iu = np.triu_indices(7)
    for ei in range(e):
        Hjj, gj, n = og.edge_blocks(t, q, s, z["Xs"], z["Cs"], int(z["ii"][ei]), int(z["jj"][ei]), z["idx"][ei],
                                    z["valid"][ei], z["Q"][ei], 1.0, 0.0, 1.5, 2, calib)
        assert bl[ei, 35] == n
        if n:
            assert np.abs(bl[ei, :28] - Hjj[iu]).max() <= 1e-4 * np.abs(Hjj).max()
            assert np.abs(bl[ei, 28:35] - gj).max() <= 1e-4 * np.abs(gj).max()
    assert not np.allclose(blocks[:, :28], bl[:, :28])


def test_config5_scale_blocks_and_solve(dev):
    """BASELINE configs[4] at reduced scale: 32 keyframes x 65 536 points, each keyframe linked to its previous
    three (slam.py:302-303) in both directions (186 directed edges).  Checks the per-edge blocks against the
    float64 oracle on a sample of edges, size-independent properties (edge-order equivariance, bitwise
    reproducibility) and a 2-iteration on-device solve (217-dim Cholesky) against the float64 oracle."""
    K_, P_ = 32, 65536
    Twc, Xs, Cs, ii, jj, idx, valid, Q = synthetic.gn_graph(K_, P_, 0, seed=17, chain=True, pose_noise=0.0)
    rng = np.random.default_rng(0)
    noisy = Twc.copy()
    noisy[1:, :3] += rng.normal(size=(K_ - 1, 3)).astype(np.float32) * 0.01
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = [t(a) for a in (noisy, Xs, Cs, ii, jj, idx, valid, Q)]
    blocks = kernels.gn_rays_blocks(*args).cpu().numpy()
    assert blocks.shape == (len(ii), 36)
    tt = noisy[:, :3].astype(np.float64); qq = noisy[:, 3:7].astype(np.float64); ss = noisy[:, 7].astype(np.float64)
    iu = np.triu_indices(7)
    for e in (0, 57, len(ii) - 1):
        Hjj, gj, n = og.edge_blocks(tt, qq, ss, Xs, Cs, int(ii[e]), int(jj[e]), idx[e], valid[e], Q[e])
        assert blocks[e, 35] == n
        # float32 per-point terms (pose, residual, Huber weight) summed over 45k points in float64
        assert np.abs(blocks[e, :28] - Hjj[iu]).max() <= 2e-4 * np.abs(Hjj).max()
        assert np.abs(blocks[e, 28:35] - gj).max() <= 2e-3 * np.abs(gj).max() + 1e-3
    # equivariance: permuting the edge list permutes the blocks, bit for bit
    perm = rng.permutation(len(ii))
    pa = [args[0], args[1], args[2]] + [t(a[perm]) for a in (ii, jj, idx, valid, Q)]
    assert np.array_equal(kernels.gn_rays_blocks(*pa).cpu().numpy(), blocks[perm])
    assert np.array_equal(kernels.gn_rays_blocks(*args).cpu().numpy(), blocks)       # reproducible
    out, info = kernels.gauss_newton_rays(*args, max_iter=2, return_info=True)
    out = out.cpu().numpy()
    assert not info["failed"] and info["iters"] == 2
    assert np.array_equal(out[0], noisy[0])                                           # pinned keyframe
    ref = og.gauss_newton_rays(noisy, Xs, Cs, ii, jj, idx, valid, Q, max_iter=2)       # float64 oracle, same graph
    assert np.isfinite(out).all() and np.abs(out - ref).max() < 5e-4


def test_edge_sharded_solve_single_rank_group(dev):
    """The edge-sharded solve (config 5: blocks per rank, all-gather of 36 doubles per edge, replicated
    assembly + solve, rank 0's step broadcast) through a ONE-rank RCCL group equals the unsharded device solve."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        args = synthetic.gn_graph(6, 2000, num_edges=11, seed=9)[:8]
        dargs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in args]
        ref = kernels.gauss_newton_rays(*dargs, max_iter=3)
        got, info = kernels.gauss_newton_rays(*dargs, max_iter=3, group=dist.group.WORLD, return_info=True)
        assert not info["failed"] and info["iters"] >= 1
        assert np.abs(got.cpu().numpy() - ref.cpu().numpy()).max() < 1e-5
    finally:
        if created:
            dist.destroy_process_group()


def test_solve_is_bitwise_reproducible(dev):
    """The dense system is assembled by a fixed-order gather (no atomics): the same call gives the same bits,
    on the single-workgroup path and on the large-graph path."""
    for kf, edges in ((8, 20), (70, 200)):
        args = synthetic.gn_graph(kf, 1500, num_edges=edges, seed=3)[:8]
        dargs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in args]
        outs = [kernels.gauss_newton_rays(*dargs, max_iter=3).cpu() for _ in range(3)]
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("n", [7, 64, 65, 128, 130, 449, 1785])
def test_blocked_cholesky_solve_any_size(dev, n):
    """kernels.cholesky_solve (linalg.py:17-50) through m3_chol_solve: blocked float64 Cholesky, block 64, for sizes
    below / at / across block boundaries and at BASELINE configs[4]'s 256 keyframes (7 * 255 = 1785 unknowns)."""
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n))
    H = A @ A.T / n + np.eye(n) * 0.5
    g = rng.normal(size=n)
    x = kernels.cholesky_solve(H, g, 1e-6)
    ref = np.linalg.solve(H + 1e-6 * np.eye(n), g)
    assert np.abs(x - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    if n == 65:
        with pytest.raises(RuntimeError, match="not positive definite"):
            kernels.cholesky_solve(-H, g, 0.0)
        xb = kernels.cholesky_solve(np.stack([H, 2 * H]), np.stack([g, g]), 0.0)          # batched form
        assert np.allclose(xb[1], 0.5 * xb[0], rtol=1e-9, atol=1e-12)


def test_blocked_cholesky_on_an_ill_conditioned_system(dev):
    """Eigenvalues spread over 1e-3 .. 1e+6 (the spread of a Gauss-Newton normal matrix with its scale column: DESIGN
    section 9 shows entries of 1e+1 beside a diagonal of 1e+8): the factorisation takes its 1 / sqrt(pivot) from
    v_rsq_f64 + Goldschmidt steps and its trsm from an explicitly inverted 64 x 64 factor - the backward error must stay
    at the float64 level, i.e. |H x - g| small against |H| |x|."""
    n = 300
    rng = np.random.default_rng(77)
    Qm, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = 10.0 ** rng.uniform(-3, 6, size=n)
    H = (Qm * lam) @ Qm.T
    H = 0.5 * (H + H.T)
    g = rng.normal(size=n)
    x = kernels.cholesky_solve(H, g, 0.0)
    resid = np.abs(H @ x - g).max()
    assert resid <= 1e-12 * np.abs(H).max() * np.abs(x).max() * n
    ref = np.linalg.solve(H, g)
    assert np.abs(x - ref).max() <= 1e-6 * np.abs(ref).max()          # condition number 1e9 x float64 epsilon


def test_blocked_cholesky_under_a_saturated_gpu(dev):
    """Round-2 advisor finding: k_chol_panel's workgroups all read the diagonal block A_kk while workgroup 0 used to write
    L_kk over it - harmless on an idle GPU (every workgroup loads within microseconds), wrong when a panel workgroup is
    dispatched late because another stream keeps the CUs busy (the intended deployment: tracking / inference beside the
    backend).  The factor now goes to a separate buffer; this runs the 1785-unknown solve several times while a second
    stream saturates the chip, and every solve must equal the float64 reference."""
    n = 1785
    rng = np.random.default_rng(5)
    A = rng.normal(size=(n, n))
    H = A @ A.T / n + np.eye(n) * 0.5
    g = rng.normal(size=n)
    ref = np.linalg.solve(H + 1e-6 * np.eye(n), g)
    Hd, gd = torch.from_numpy(H).to(dev), torch.from_numpy(g).to(dev)
    side = torch.cuda.Stream()
    x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    junk = torch.empty(64 * 1024 * 1024, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):                                   # ~0.3 s of work that occupies every CU: matrix + streaming kernels
        for _ in range(60):
            y = x @ x
            junk.add_(1.0)
    outs = [kernels.cholesky_solve(Hd, gd, 1e-6) for _ in range(6)]   # on the current stream, beside the load
    busy = not side.query()
    torch.cuda.synchronize()
    assert busy, "the side stream finished before the solves were issued: the test did not overlap anything"
    for o in outs:
        assert np.abs(o.cpu().numpy() - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    del y
    # float32 in -> float32 out (linalg.py:17-50 returns H.dtype), arithmetic still float64
    x32 = kernels.cholesky_solve(H[:65, :65].astype(np.float32), g[:65].astype(np.float32), 1e-6)
    assert x32.dtype == np.float32
    r32 = np.linalg.solve(H[:65, :65].astype(np.float32).astype(np.float64) + 1e-6 * np.eye(65), g[:65].astype(np.float32).astype(np.float64))
    assert np.abs(x32 - r32).max() <= 1e-6 * max(1.0, np.abs(r32).max())


def test_blocked_cholesky_more_block_columns_than_compute_units(dev):
    """Round-3 advisor finding: the backward substitution is one launch of flag-chained workgroups (consumer index >
    producer index).  320 block columns (20 480 unknowns) are more workgroups than the chip has CUs (256), so late
    workgroups are dispatched while earlier ones already spin on their flags; the solve must finish (the spin is
    bounded: a stuck chain would come back as `not positive definite`, never hang) and reach a float64 residual."""
    n = 64 * 320
    g = torch.Generator(device="cpu").manual_seed(9)
    R = torch.randn(n, n, generator=g, dtype=torch.float32).to(dev).double() * 0.005
    H = 0.5 * (R + R.T)
    del R
    H.diagonal().add_(200.0)                                         # strictly diagonally dominant: |off-diagonal row sum| ~ 80
    b = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    x = kernels.cholesky_solve(H, b, 0.0)
    resid = float((H @ x - b).abs().max())
    assert resid <= 1e-10 * float(b.abs().max())
    assert x.dtype == torch.float64 and bool(torch.isfinite(x).all())


def test_large_graph_solve_stays_on_the_device(dev):
    """70 keyframes -> 483 unknowns (beyond the 63 a single workgroup factors in place: blocked Cholesky): the whole Gauss-Newton loop (blocks, assembly,
    blocked Cholesky, stop test, retraction) runs as one stream-ordered call and matches the float64 oracle; a
    converged solve stops by its device flag."""
    K_, P_ = 70, 4096
    Twc, Xs, Cs, ii, jj, idx, valid, Q = synthetic.gn_graph(K_, P_, 0, seed=23, chain=True, pose_noise=0.0)
    rng = np.random.default_rng(1)
    noisy = Twc.copy()
    noisy[1:, :3] += rng.normal(size=(K_ - 1, 3)).astype(np.float32) * 0.01
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = [t(a) for a in (noisy, Xs, Cs, ii, jj, idx, valid, Q)]
    out, info = kernels.gauss_newton_rays(*args, max_iter=3, return_info=True)
    ref = og.gauss_newton_rays(noisy, Xs, Cs, ii, jj, idx, valid, Q, max_iter=3)
    assert not info["failed"] and info["iters"] >= 1
    assert np.array_equal(out[0].cpu().numpy(), noisy[0]) and np.abs(out.cpu().numpy() - ref).max() < 5e-4
    # the device stop flag: a threshold above the first step's norm stops BEFORE the update (gauss_newton.py:262-265)
    out2, info2 = kernels.gauss_newton_rays(*args, max_iter=5, delta_thresh=1e6, return_info=True)
    assert info2["stopped"] and not info2["failed"] and info2["iters"] == 0
    assert np.array_equal(out2.cpu().numpy(), noisy)
