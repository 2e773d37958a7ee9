"""CPU: the oracle (our restatement) against golden vectors frozen from the reference's
importable numpy twins (oracle/make_golden.py).  These tests PIN the oracle."""
import os

import numpy as np

from oracle import gn_rays as og
from oracle import matching as om
from oracle import sim3 as S


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_iter_proj_matches_reference_bit_exact(golden_dir):
    for tag in ("b1", "b2", "earlystop"):
        z = _load(golden_dir, f"iter_proj_{tag}.npz")
        p, v = om.iter_proj(z["rays_with_grad"], z["pts3d_norm"], z["p_init"], int(z["max_iter"]),
                            float(z["lambda_init"]), float(z["convergence_thresh"]), "global")
        assert np.array_equal(p, z["p_ref"]), tag          # float32 positions, bit for bit
        assert np.array_equal(v, z["valid_ref"]), tag
        assert v.mean() > 0.5


EDGE_ITER = ("edge_it1", "edge_it3", "edge_lam", "refbench")
EDGE_REFINE = ("r2_d16", "r4_d24", "r1_d64", "r3_d5")


def test_iter_proj_edge_cases_match_reference_bit_exact(golden_dir):
    """Reference-generated (round 4): a random ray map (steps that leave the image, the determinant clamp), points starting
    outside the image / on its corners, N != H * W, 1 / 3 / 10 iterations, lambda 1e-2."""
    for tag in EDGE_ITER:
        z = _load(golden_dir, f"iter_proj_{tag}.npz")
        p, v = om.iter_proj(z["rays_with_grad"], z["pts3d_norm"], z["p_init"], int(z["max_iter"]),
                            float(z["lambda_init"]), float(z["convergence_thresh"]), "global")
        assert np.array_equal(p, z["p_ref"]), tag
        assert np.array_equal(v, z["valid_ref"]), tag
        assert 0.02 < v.mean() < 0.98, tag                  # both outcomes of the validity test occur


def test_refine_matches_other_lengths_and_radii_bit_exact(golden_dir):
    """Reference-generated (round 4): D = 16 / 24 / 64 / 5, radius 2 / 4 / 1 / 3, random descriptors, centres beyond every
    border, a constant patch (ties: the first candidate in raster order wins)."""
    for tag in EDGE_REFINE:
        z = _load(golden_dir, f"refine_matches_{tag}.npz")
        r = om.refine_matches(z["D11"], z["D21"], z["p1"], int(z["radius"]), int(z["dilation_max"]))
        assert np.array_equal(r, z["p_ref"]), tag


def test_iter_proj_early_stop_is_exercised(golden_dir):
    z = _load(golden_dir, "iter_proj_earlystop.npz")
    args = (z["rays_with_grad"], z["pts3d_norm"], z["p_init"], int(z["max_iter"]), float(z["lambda_init"]))
    p_stop, _ = om.iter_proj(*args, float(z["convergence_thresh"]), "global")
    p_full, _ = om.iter_proj(*args, 0.0, "global")
    assert not np.array_equal(p_stop, p_full)              # the threshold really cut the iteration short
    assert np.array_equal(p_stop, z["p_ref"])


def test_iter_proj_batch_scope_equals_loop_of_single_calls(golden_dir):
    z = _load(golden_dir, "iter_proj_b2.npz")
    pb, vb = om.iter_proj(z["rays_with_grad"], z["pts3d_norm"], z["p_init"], 10, 1e-8, 1e-6, "batch")
    for b in range(2):
        p1, v1 = om.iter_proj(z["rays_with_grad"][b:b + 1], z["pts3d_norm"][b:b + 1], z["p_init"][b:b + 1],
                              10, 1e-8, 1e-6, "global")
        assert np.array_equal(pb[b], p1[0]) and np.array_equal(vb[b], v1[0])


def test_refine_matches_bit_exact(golden_dir):
    for dmax in (0, 2):
        z = _load(golden_dir, f"refine_matches_d{dmax}.npz")
        r = om.refine_matches(z["D11"], z["D21"], z["p1"], int(z["radius"]), int(z["dilation_max"]))
        assert np.array_equal(r, z["p_ref"])
        assert (r != z["p1"]).any(-1).mean() > 0.5          # the search moved most points
    # numpy-twin semantics: dilation_max has no effect; chained (Metal) semantics differ
    z = _load(golden_dir, "refine_matches_d2.npz")
    c = om.refine_matches(z["D11"], z["D21"], z["p1"], 3, 2, chained=True)
    assert not np.array_equal(c, z["p_ref"])
    assert np.abs(c - z["p1"]).max() <= 3 * 2 + 3


def test_gauss_newton_rays_matches_reference(golden_dir):
    for tag, tol in (("it1", 1e-6), ("it3", 1e-6), ("chain", 1e-6)):
        z = _load(golden_dir, f"gn_rays_{tag}.npz")
        out = og.gauss_newton_rays(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                   max_iter=int(z["max_iter"]), pin=int(z["pin"]))
        assert np.abs(out - z["Twc_ref"]).max() <= tol, tag
        assert np.abs(z["Twc_ref"] - z["Twc"]).max() > 1e-3      # the solve moved the poses
        assert np.array_equal(out[0], z["Twc"][np.unique(np.concatenate([z["ii"], z["jj"]]))[0]])  # pinned


def test_gauss_newton_with_every_argument_off_its_default(golden_dir):
    """Reference-generated (round 4): sigma, C_thresh and Q_thresh that cut points, pin = 2, a delta_thresh that stops the
    loop early, a repeated and a reversed edge - rays and points variants."""
    z = _load(golden_dir, "gn_rays_params.npz")
    kw = {k: (int(z[k]) if k in ("max_iter", "pin") else float(z[k])) for k in
          ("sigma_ray", "sigma_dist", "C_thresh", "Q_thresh", "max_iter", "delta_thresh", "pin")}
    out = og.gauss_newton_rays(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"], **kw)
    assert np.abs(out - z["Twc_ref"]).max() <= 1e-6 and np.abs(z["Twc_ref"] - z["Twc"]).max() > 1e-2
    assert np.array_equal(out[:2], z["Twc"][:2])                     # two pinned poses
    z = _load(golden_dir, "gn_points_params.npz")
    kw = {k: (int(z[k]) if k in ("max_iter", "pin") else float(z[k])) for k in
          ("sigma_point", "C_thresh", "Q_thresh", "max_iter", "delta_thresh", "pin")}
    out = og.gauss_newton_points(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"], **kw)
    assert np.abs(out - z["Twc_ref"]).max() <= 1e-6 and np.array_equal(out[:2], z["Twc"][:2])


def test_gauss_newton_calib_with_every_argument_off_its_default(golden_dir):
    z = _load(golden_dir, "gn_calib_params.npz")
    kw = {k: (int(z[k]) if k in ("max_iter", "pin", "pixel_border") else float(z[k])) for k in
          ("pixel_border", "z_eps", "sigma_pixel", "sigma_depth", "C_thresh", "Q_thresh", "max_iter", "delta_thresh", "pin")}
    out = og.gauss_newton_calib(z["Twc"], z["Xs"], z["Cs"], z["K"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                tuple(int(v) for v in z["img_size"]), **kw)
    assert np.abs(out - z["Twc_ref"]).max() <= 1e-6 and np.abs(z["Twc_ref"] - z["Twc"]).max() > 1e-3
    assert np.array_equal(out[:2], z["Twc"][:2])


def test_sim3_ops_known_answers(golden_dir):
    z = _load(golden_dir, "sim3_ops.npz")
    assert np.array_equal(S.quat_multiply(z["q1"], z["q2"]), z["qmul"])
    assert np.array_equal(S.quat_rotate(z["q1"], z["v"]), z["qrot"])
    for a, b in zip(S.sim3_relative(z["t1"], z["q1"], z["s1"], z["t2"], z["q2"], z["s2"]),
                    (z["rel_t"], z["rel_q"], z["rel_s"])):
        assert np.array_equal(a, b)
    assert np.array_equal(S.exp_so3(z["xi"][:, 3:6]), z["exp_so3"])
    for a, b in zip(S.exp_sim3(z["xi"]), (z["exp_t"], z["exp_q"], z["exp_s"])):
        assert np.array_equal(a, b)
    for a, b in zip(S.retract_sim3(z["xi"], z["t1"], z["q1"], z["s1"]), (z["retr_t"], z["retr_q"], z["retr_s"])):
        assert np.array_equal(a, b)
    assert np.array_equal(S.huber_weight(z["hub_r"]), z["hub_w"])


def test_sim3_ops_remaining_helpers(golden_dir):
    """quat_inv, sim3_act (batched, broadcast over points) and huber_weight with other k: exact against the reference."""
    z = _load(golden_dir, "sim3_ops_more.npz")
    assert np.array_equal(S.quat_inv(z["q"]), z["qinv"])
    act = S.sim3_act(z["t"][:, None, :], z["q"][:, None, :], z["s"][:, None, 0], z["X"])
    assert np.array_equal(act, z["act"])
    assert np.array_equal(S.huber_weight(z["r"], 2.0), z["hub_k2"]) and np.array_equal(S.huber_weight(z["r"], 0.5), z["hub_k05"])


def test_cholesky_solve_backend_sized_system(golden_dir):
    z = _load(golden_dir, "cholesky_solve_n210.npz")
    x = og.cholesky_solve(z["H"], z["g"])
    assert np.abs(x - z["x"]).max() <= 1e-9 * np.abs(z["x"]).max()


def test_cholesky_solve(golden_dir):
    z = _load(golden_dir, "cholesky_solve.npz")
    assert np.allclose(og.cholesky_solve(z["H"], z["g"]), z["x"], rtol=0, atol=1e-12)
    assert np.allclose(og.cholesky_solve(z["H32"], z["g32"]), z["x32"], rtol=1e-5, atol=1e-6)


def test_gauss_newton_points_matches_reference(golden_dir):
    for tag in ("it1", "it3"):
        z = _load(golden_dir, f"gn_points_{tag}.npz")
        out = og.gauss_newton_points(z["Twc"], z["Xs"], z["Cs"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                     max_iter=int(z["max_iter"]), pin=int(z["pin"]))
        assert np.abs(out - z["Twc_ref"]).max() <= 1e-6, tag
        assert np.abs(z["Twc_ref"] - z["Twc"]).max() > 1e-3


def test_gauss_newton_calib_matches_reference(golden_dir):
    for tag in ("it1", "it3"):
        z = _load(golden_dir, f"gn_calib_{tag}.npz")
        out = og.gauss_newton_calib(z["Twc"], z["Xs"], z["Cs"], z["K"], z["ii"], z["jj"], z["idx"], z["valid"], z["Q"],
                                    tuple(int(v) for v in z["img_size"]), max_iter=int(z["max_iter"]), pin=int(z["pin"]))
        ref = z["Twc_ref"]
        assert np.isfinite(ref).all()
        assert np.abs(out - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), tag
        assert np.abs(ref - z["Twc"]).max() > 1e-3
