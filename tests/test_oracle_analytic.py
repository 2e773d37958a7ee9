"""CPU: known-answer tests for the parts of the oracle that have NO executable reference
(tracker GN restated from MLX source text): finite-difference Jacobians and recovery of
a known Sim(3)."""
import numpy as np

from mast3r_slam import synthetic
from oracle import sim3 as S
from oracle import tracking as ot


def test_point_to_ray_dist_jacobian_fd():
    rng = np.random.default_rng(0)
    X = rng.normal(size=(50, 3)) + np.array([0, 0, 3.0])
    rd, J = ot.point_to_ray_dist(X, jacobian=True)
    eps = 1e-6
    for k in range(3):
        d = np.zeros(3); d[k] = eps
        fd = (ot.point_to_ray_dist(X + d) - ot.point_to_ray_dist(X - d)) / (2 * eps)
        assert np.allclose(J[:, :, k], fd, atol=1e-7)
    assert np.allclose(np.linalg.norm(rd[:, :3], axis=-1), 1.0, atol=1e-9)


def test_act_sim3_jacobian_is_left_perturbation():
    """geometry.py:118-137: J = [I, -[pW]x, pW] is d/dtau of exp(tau) * T acting on p (tau -> 0)."""
    rng = np.random.default_rng(1)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    T = np.concatenate([rng.normal(size=3), q, [1.3]])
    p = rng.normal(size=(20, 3))
    pW, J = ot.act_sim3(T, p, jacobian=True)
    eps = 1e-6
    for k in range(7):
        tau = np.zeros(7); tau[k] = eps
        Tp = S.sim3_mul_mlx(S.sim3_exp_mlx(tau), T)
        Tm = S.sim3_mul_mlx(S.sim3_exp_mlx(-tau), T)
        fd = (S.sim3_act_mlx(Tp, p) - S.sim3_act_mlx(Tm, p)) / (2 * eps)
        assert np.allclose(J[:, :, k], fd, atol=1e-6), k


def test_sim3_group_axioms():
    rng = np.random.default_rng(2)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    T = np.concatenate([rng.normal(size=3), q, [0.8]])
    I = S.sim3_mul_mlx(T, S.sim3_inv_mlx(T))
    assert np.allclose(I, [0, 0, 0, 0, 0, 0, 1, 1], atol=1e-9)
    p = rng.normal(size=(5, 3))
    assert np.allclose(S.sim3_act_mlx(S.sim3_inv_mlx(T), S.sim3_act_mlx(T, p)), p, atol=1e-9)
    assert np.allclose(S.sim3_exp_mlx(np.zeros(7)), [0, 0, 0, 0, 0, 0, 1, 1], atol=1e-12)


def test_tracking_recovers_known_sim3():
    pr = synthetic.tracking_problem(32, 40, seed=3, noise=0.0, perm_frac=0.0, valid_p=1.0)
    Xf = pr["Xf_canon"][pr["idx"]]
    T_WCf, T_rel, info = ot.opt_pose_ray_dist_sim3(Xf, pr["Xk"], pr["T_WCf"], pr["T_WCk"], pr["Qk"] + 1.0,
                                                  pr["valid"], cfg=dict(max_iters=50, rel_error=0, delta_norm=1e-9))
    # right-multiplied retraction with a left-perturbation Jacobian still converges (slower) for small motion
    assert np.allclose(T_rel[:3], pr["T_true"][:3], atol=2e-4)
    assert np.allclose(np.abs(T_rel[3:7]), np.abs(pr["T_true"][3:7]), atol=2e-4)
    assert abs(T_rel[7] - pr["T_true"][7]) < 2e-4
    assert info["costs"][-1] < 1e-3 * info["costs"][0]


def test_tracking_default_config_converges_and_reports():
    pr = synthetic.tracking_problem(24, 32, seed=5)
    Xf = pr["Xf_canon"][pr["idx"]]
    _, T_rel, info = ot.opt_pose_ray_dist_sim3(Xf, pr["Xk"], pr["T_WCf"], pr["T_WCk"], pr["Qk"], pr["valid"])
    assert 1 <= info["iters"] <= 10
    assert info["costs"][-1] < info["costs"][0]
    assert np.linalg.norm(T_rel[:3] - pr["T_true"][:3]) < 0.02


def test_validity_and_match_quality():
    Qff = np.array([4.0, 1.0, 9.0]); Qkf = np.array([1.0, 4.0, 0.25]); idx = np.array([2, 0, 1])
    Qk = ot.match_quality(Qff, Qkf, idx)
    assert np.allclose(Qk, [3.0, 4.0, 0.5])
    vo, vk = ot.validity(np.array([True, True, True]), np.array([1.0, -1.0, 1.0]), np.ones(3), Qk)
    assert vo.tolist() == [True, False, False] and vk.tolist() == [True, True, False]


def test_project_calib_jacobian_fd_and_gates():
    rng = np.random.default_rng(4)
    K = np.array([[400.0, 0, 160], [0, 420.0, 120], [0, 0, 1]])
    P = rng.normal(size=(40, 3)) * 0.3 + np.array([0, 0, 3.0])
    pz, J, valid = ot.project_calib(P, K, (240, 320), jacobian=True)
    eps = 1e-6
    for k in range(3):
        d = np.zeros(3); d[k] = eps
        fd = (ot.project_calib(P + d, K, (240, 320))[0] - ot.project_calib(P - d, K, (240, 320))[0]) / (2 * eps)
        assert np.allclose(J[:, :, k], fd, atol=1e-5)
    assert valid.all()
    Pb = np.array([[0.0, 0.0, -1.0], [50.0, 0.0, 1.0]])
    pz2, v2 = ot.project_calib(Pb, K, (240, 320))
    assert not v2.any() and pz2[0, 2] == 0.0                              # behind the camera / outside the image


def test_calibrated_tracking_recovers_known_sim3():
    h, w = 24, 32
    K = np.array([[float(w), 0, w / 2], [0, float(w), h / 2], [0, 0, 1]])
    pr = synthetic.tracking_problem(h, w, seed=6, noise=0.0, perm_frac=0.0, valid_p=1.0)
    Xk = ot.constrain_points_to_ray((h, w), pr["Xk"].astype(np.float64), K)
    assert np.allclose(Xk, pr["Xk"], atol=1e-6)                            # the synthetic surface is back-projected with f = w
    Xf = pr["Xf_canon"][pr["idx"]]
    _, T_rel, info = ot.opt_pose_calib_sim3(Xf, Xk, pr["T_WCf"], pr["T_WCk"], pr["Qk"] + 1.0, pr["valid"], K, (h, w),
                                            cfg=dict(max_iters=60, rel_error=0, delta_norm=1e-10))
    assert np.allclose(T_rel[:3], pr["T_true"][:3], atol=5e-4)
    assert abs(T_rel[7] - pr["T_true"][7]) < 5e-4
    assert info["costs"][-1] < 1e-4 * info["costs"][0]


def test_fast_reciprocal_nn_oracle_known_answers():
    """Own-semantics op: identical maps give the identity on every seed; a permuted copy gives the
    permutation; duplicates resolve to the lowest index."""
    import numpy as np
    from oracle import matching as om
    rng = np.random.default_rng(0)
    D = rng.normal(size=(16, 24, 24)).astype(np.float32)
    D /= np.linalg.norm(D, axis=-1, keepdims=True)
    i1, i2 = om.fast_reciprocal_nn(D, D, subsample=4)
    assert len(i1) == 4 * 6 and np.array_equal(i1, i2)
    perm = rng.permutation(16 * 24)
    D2 = D.reshape(-1, 24)[perm].reshape(16, 24, 24)                    # D2[j] = D[perm[j]]
    i1, i2 = om.fast_reciprocal_nn(D, D2, subsample=4)
    assert np.array_equal(perm[i2], i1)
    idx, best, second = om.nn_search(D.reshape(-1, 24)[:5], np.concatenate([D.reshape(-1, 24), D.reshape(-1, 24)]))
    assert idx.tolist() == [0, 1, 2, 3, 4] and np.allclose(best, 1.0) and np.allclose(second, 1.0)
