"""GPU parity: fused Gauss-Newton tracking solve vs the float64 CPU oracle.
Floating point: HIP computes per-point terms in float32 and accumulates in float64;
tolerances are stated per assertion."""
import numpy as np
import pytest
import torch

from mast3r_slam import synthetic, tracker
from oracle import sim3 as S
from oracle import tracking as ot

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _problem(h, w, seed, **kw):
    pr = synthetic.tracking_problem(h, w, seed=seed, **kw)
    pr["Xf"] = pr["Xf_canon"][pr["idx"]]
    return pr


def test_normal_equations_match_oracle(dev):
    pr = _problem(48, 64, 1)
    rng = np.random.default_rng(0)
    q = np.array([0.01, -0.02, 0.015, 1.0]); q /= np.linalg.norm(q)
    T = np.concatenate([rng.normal(size=3) * 0.02, q, [1.01]]).astype(np.float32)
    H, g, cost = tracker.normal_equations(_t(pr["Xf"], dev), _t(pr["Xk"], dev), _t(T, dev), _t(pr["Qk"], dev),
                                          _t(pr["valid"], dev))
    # oracle: one _solve at the same pose
    Xf = pr["Xf"].astype(np.float64); Xk = pr["Xk"].astype(np.float64)
    v = pr["valid"].astype(np.float64)[:, None]; Qk = pr["Qk"].astype(np.float64)[:, None]
    si = np.concatenate([np.repeat(v * np.sqrt(Qk) / 0.003, 3, 1), v * np.sqrt(Qk) / 10.0], 1)
    p, dT = ot.act_sim3(T.astype(np.float64), Xf, True)
    rd, drd = ot.point_to_ray_dist(p, True)
    _, cost_o, H_o, g_o = ot.solve_step(si, ot.point_to_ray_dist(Xk) - rd, -drd @ dT)
    Hs = np.abs(H_o).max()
    assert np.abs(H.cpu().numpy() - H_o).max() <= 2e-5 * Hs              # float32 per-point terms
    assert np.abs(g.cpu().numpy() - g_o).max() <= 2e-5 * np.abs(g_o).max()
    assert abs(float(cost) - cost_o) <= 2e-5 * cost_o
    assert np.allclose(H.cpu().numpy(), H.cpu().numpy().T)


@pytest.mark.parametrize("fixed", [False, True])
def test_gn_loop_matches_oracle(dev, fixed):
    pr = _problem(48, 64, 2)
    Tf, Trel, info = tracker.opt_pose_ray_dist_sim3(_t(pr["Xf"], dev), _t(pr["Xk"], dev), _t(pr["T_WCf"], dev),
                                                    _t(pr["T_WCk"], dev), _t(pr["Qk"], dev), _t(pr["valid"], dev),
                                                    fixed_iters=fixed)
    To, Trel_o, io = ot.opt_pose_ray_dist_sim3(pr["Xf"], pr["Xk"], pr["T_WCf"], pr["T_WCk"], pr["Qk"], pr["valid"],
                                               fixed_iters=10 if fixed else None)
    info = info.cpu().numpy()
    assert int(info[0]) == io["iters"]                                    # same stopping iteration
    assert np.abs(Trel.cpu().numpy() - Trel_o).max() < 5e-5               # pose: abs 5e-5 (t, q, s ~ O(1))
    assert np.abs(Tf.cpu().numpy() - To).max() < 5e-5
    assert abs(info[1] - io["costs"][-1]) <= 1e-3 * io["costs"][-1]
    assert bool(info[3]) == (not fixed)


def test_nonidentity_world_poses(dev):
    pr = _problem(32, 48, 3)
    rng = np.random.default_rng(5)
    qk = rng.normal(size=4); qk /= np.linalg.norm(qk)
    T_WCk = np.concatenate([rng.normal(size=3), qk, [1.7]]).astype(np.float32)
    T_WCf = S.sim3_mul_mlx(T_WCk.astype(np.float64), np.array([0.01, 0, 0, 0, 0, 0, 1, 1.0])).astype(np.float32)
    Tf, Trel, _ = tracker.opt_pose_ray_dist_sim3(_t(pr["Xf"], dev), _t(pr["Xk"], dev), _t(T_WCf, dev),
                                                 _t(T_WCk, dev), _t(pr["Qk"], dev), _t(pr["valid"], dev))
    To, Trel_o, _ = ot.opt_pose_ray_dist_sim3(pr["Xf"], pr["Xk"], T_WCf, T_WCk, pr["Qk"], pr["valid"])
    assert np.abs(Trel.cpu().numpy() - Trel_o).max() < 5e-5
    assert np.abs(Tf.cpu().numpy() - To).max() < 2e-4                     # scaled by |T_WCk| ~ 2


def test_full_size_recovers_known_sim3_and_is_deterministic(dev):
    pr = _problem(512, 512, 0)
    a = [_t(pr[k], dev) for k in ("Xf", "Xk", "T_WCf", "T_WCk", "Qk", "valid")]
    _, T1, i1 = tracker.opt_pose_ray_dist_sim3(*a, cfg=dict(max_iters=30, rel_error=0.0, delta_norm=1e-7))
    _, T2, i2 = tracker.opt_pose_ray_dist_sim3(*a, cfg=dict(max_iters=30, rel_error=0.0, delta_norm=1e-7))
    assert torch.equal(T1, T2) and torch.equal(i1, i2)                    # fixed-order reduction: bitwise
    T = T1.cpu().numpy()
    assert np.abs(T[:3] - pr["T_true"][:3]).max() < 2e-3                  # 5 % outlier matches + noise 1e-3
    assert abs(T[7] - pr["T_true"][7]) < 2e-3
    assert np.abs(np.abs(T[3:7]) - np.abs(pr["T_true"][3:7])).max() < 1e-3


@pytest.mark.parametrize("fixed", [False, True])
def test_full_size_512_solve_matches_the_oracle_loop(dev, fixed):
    """BASELINE configs[2] at its own size (262 144 points, the size bench.py times): the device loop stops at the same
    iteration as oracle.tracking.opt_pose_ray_dist_sim3 (tracker.py:258-324, float64) and returns its pose - relative
    pose and frame pose within 5e-5 absolute, final cost within 1e-3 relative - with the convergence test on
    (reference behaviour) and with the bench's 10 fixed iterations."""
    pr = _problem(512, 512, 4)
    a = [_t(pr[k], dev) for k in ("Xf", "Xk", "T_WCf", "T_WCk", "Qk", "valid")]
    Tf, Trel, info = tracker.opt_pose_ray_dist_sim3(*a, fixed_iters=fixed)
    To, Trel_o, io = ot.opt_pose_ray_dist_sim3(pr["Xf"], pr["Xk"], pr["T_WCf"], pr["T_WCk"], pr["Qk"], pr["valid"],
                                               fixed_iters=10 if fixed else None)
    info = info.cpu().numpy()
    assert int(info[0]) == io["iters"] and (fixed or io["iters"] < 10)
    assert np.abs(Trel.cpu().numpy() - Trel_o).max() < 5e-5
    assert np.abs(Tf.cpu().numpy() - To).max() < 5e-5
    assert abs(info[1] - io["costs"][-1]) <= 1e-3 * io["costs"][-1]
    # batch of the bench's shape: 8 problems per launch sequence, problem 3 is this one
    others = [_problem(512, 512, 40 + i) for i in range(2)]
    st = lambda k: torch.stack([_t(p[k], dev) for p in (others[0], pr, others[1])])
    Tfb, Trelb, infob = tracker.opt_pose_ray_dist_sim3(st("Xf"), st("Xk"), st("T_WCf"), st("T_WCk"), st("Qk"), st("valid"),
                                                       fixed_iters=fixed)
    assert torch.equal(Trelb[1], Trel) and torch.equal(Tfb[1], Tf) and torch.equal(infob[1].cpu(), torch.from_numpy(info))


@pytest.mark.parametrize("iters", [0, 1, 2, 3])
def test_iteration_budgets_across_the_double_buffered_state(dev, iters):
    """One launch per iteration: the step that follows an accumulation runs in the prologue of the NEXT launch and the
    state / partial buffers alternate by parity - budgets of 0 (no launch at all: the initial relative pose comes back),
    1 (launch 0 + the stand-alone last step), 2 and 3 (odd / even final slots) must all agree with the oracle loop."""
    pr = _problem(40, 56, 11)
    cfg = dict(max_iters=iters)
    Tf, Trel, info = tracker.opt_pose_ray_dist_sim3(_t(pr["Xf"], dev), _t(pr["Xk"], dev), _t(pr["T_WCf"], dev),
                                                    _t(pr["T_WCk"], dev), _t(pr["Qk"], dev), _t(pr["valid"], dev), cfg,
                                                    fixed_iters=True)
    info = info.cpu().numpy()
    assert int(info[0]) == iters and info[3] == 0.0
    if iters == 0:
        T0 = S.sim3_mul_mlx(S.sim3_inv_mlx(pr["T_WCk"].astype(np.float64)), pr["T_WCf"].astype(np.float64))
        assert np.abs(Trel.cpu().numpy() - T0).max() < 1e-6
        return
    To, Trel_o, io = ot.opt_pose_ray_dist_sim3(pr["Xf"], pr["Xk"], pr["T_WCf"], pr["T_WCk"], pr["Qk"], pr["valid"],
                                               fixed_iters=iters)
    assert np.abs(Trel.cpu().numpy() - Trel_o).max() < 5e-5
    assert np.abs(Tf.cpu().numpy() - To).max() < 5e-5


def test_all_invalid_and_degenerate_input_do_not_crash(dev):
    pr = _problem(16, 16, 7)
    zeros = np.zeros_like(pr["valid"])
    Tf, Trel, info = tracker.opt_pose_ray_dist_sim3(_t(pr["Xf"], dev), _t(pr["Xk"], dev), _t(pr["T_WCf"], dev),
                                                    _t(pr["T_WCk"], dev), _t(pr["Qk"], dev), _t(zeros, dev))
    # H = 1e-6 I, g = 0 -> tau = 0 -> converged by |tau| at the first step, pose unchanged
    assert np.allclose(Trel.cpu().numpy(), [0, 0, 0, 0, 0, 0, 1, 1], atol=1e-7)
    assert int(info.cpu()[0]) == 1 and bool(info.cpu()[3])


@pytest.mark.parametrize("n", [5000, 4999, 3])      # 4 points per lane / the one-point path (N % 4 != 0) / tiny
def test_track_gather_and_sim3_act(dev, n):
    rng = np.random.default_rng(9)
    Xc = rng.normal(size=(n, 3)).astype(np.float32)
    Cf = rng.uniform(-0.5, 3, n).astype(np.float32); Ck = rng.uniform(-0.5, 3, n).astype(np.float32)
    Qff = rng.uniform(0.5, 4, n).astype(np.float32); Qkf = rng.uniform(0.5, 4, n).astype(np.float32)
    idx = rng.integers(0, n, n).astype(np.int64); vm = rng.uniform(size=n) < 0.8
    Xf, Qk, vo, vk, cnt = tracker.track_gather(*[_t(a, dev) for a in (Xc, Cf, Ck, Qff, Qkf, idx, vm)], 0.0, 1.5)
    Qk_o = np.sqrt(Qff[idx] * Qkf)
    vo_o, vk_o = ot.validity(vm, Cf[idx], Ck, Qk_o, 0.0, 1.5)
    assert np.array_equal(Xf.cpu().numpy(), Xc[idx])
    assert np.array_equal(Qk.cpu().numpy(), Qk_o)                         # one mul + correctly rounded sqrt
    assert np.array_equal(vo.cpu().numpy().astype(bool), vo_o) and np.array_equal(vk.cpu().numpy().astype(bool), vk_o)
    assert cnt.cpu().tolist() == [int(vo_o.sum()), int(vk_o.sum())]
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    T = np.concatenate([rng.normal(size=3), q, [1.2]]).astype(np.float32)
    out = tracker.sim3_act(_t(T, dev), _t(Xc, dev)).cpu().numpy()
    assert np.abs(out - S.sim3_act_mlx(T.astype(np.float64), Xc.astype(np.float64))).max() < 1e-5


def test_track_gather_tiled_path_on_coherent_matches(dev):
    """k_track_gather_lds: workgroups whose 1024 keyframe points match a compact range of frame points gather from an LDS
    copy of that range, the others (scattered indices, ranges beyond the cap, negative indices, the ragged last
    workgroup) from global memory - every output exact against numpy either way, counts included."""
    rng = np.random.default_rng(21)
    n = 512 * 40 + 1024 * 3 + 512                                       # not a multiple of the 1024-point workgroup
    Xc = rng.normal(size=(2, n, 3)).astype(np.float32)
    Cf = rng.uniform(-0.5, 3, (2, n)).astype(np.float32); Ck = rng.uniform(-0.5, 3, (2, n)).astype(np.float32)
    Qff = rng.uniform(0.5, 4, (2, n)).astype(np.float32); Qkf = rng.uniform(0.5, 4, (2, n)).astype(np.float32)
    idx = np.arange(n)[None].repeat(2, 0) + rng.integers(-700, 700, size=(2, n))     # coherent: within +-700 of the identity
    idx = np.clip(idx, 0, n - 1)
    idx[0, 4096:6144] = rng.integers(0, n, 2048)                         # two workgroups of scattered matches
    idx[1, 9000:9010] = np.arange(10) - 5 - 3                            # negative indices (wrap by + n, as the kernel does)
    vm = rng.uniform(size=(2, n)) < 0.8
    Xf, Qk, vo, vk, cnt = tracker.track_gather(*[_t(a, dev) for a in (Xc, Cf, Ck, Qff, Qkf, idx.astype(np.int64), vm)], 0.0, 1.5)
    for b in range(2):
        ii = np.where(idx[b] < 0, idx[b] + n, idx[b])
        Qk_o = np.sqrt(Qff[b][ii] * Qkf[b])
        vo_o, vk_o = ot.validity(vm[b], Cf[b][ii], Ck[b], Qk_o, 0.0, 1.5)
        assert np.array_equal(Xf[b].cpu().numpy(), Xc[b][ii])
        assert np.array_equal(Qk[b].cpu().numpy(), Qk_o)
        assert np.array_equal(vo[b].cpu().numpy().astype(bool), vo_o) and np.array_equal(vk[b].cpu().numpy().astype(bool), vk_o)
        assert cnt[b].cpu().tolist() == [int(vo_o.sum()), int(vk_o.sum())]


def test_batched_solve_equals_loop_of_single_solves(dev):
    """P problems in one launch sequence == P separate calls, bit for bit (pairs are independent units)."""
    prs = [_problem(32, 48, 20 + i) for i in range(3)]
    st = lambda k: torch.stack([_t(p[k], dev) for p in prs])
    Tf, Trel, info = tracker.opt_pose_ray_dist_sim3(st("Xf"), st("Xk"), st("T_WCf"), st("T_WCk"), st("Qk"), st("valid"))
    assert Tf.shape == (3, 8) and info.shape == (3, 4)
    for i, p in enumerate(prs):
        a = [_t(p[k], dev) for k in ("Xf", "Xk", "T_WCf", "T_WCk", "Qk", "valid")]
        Tf1, Trel1, info1 = tracker.opt_pose_ray_dist_sim3(*a)
        assert torch.equal(Tf[i], Tf1) and torch.equal(Trel[i], Trel1) and torch.equal(info[i], info1)
    # batched gather
    rng = np.random.default_rng(3)
    n = 1000
    Xc = rng.normal(size=(2, n, 3)).astype(np.float32); C = rng.uniform(0, 2, (2, n)).astype(np.float32)
    Q = rng.uniform(0.5, 4, (2, n)).astype(np.float32); idx = rng.integers(0, n, (2, n)); vm = rng.uniform(size=(2, n)) < 0.7
    Xf, Qk, vo, vk, cnt = tracker.track_gather(_t(Xc, dev), _t(C, dev), _t(C, dev), _t(Q, dev), _t(Q, dev), _t(idx, dev), _t(vm, dev))
    for b in range(2):
        assert np.array_equal(Xf[b].cpu().numpy(), Xc[b][idx[b]])
        assert cnt[b].cpu().tolist()[1] == int((vm[b] & (np.sqrt(Q[b][idx[b]] * Q[b]) > 1.5)).sum())


@pytest.mark.parametrize("h,w", [(48, 64), (48, 62), (33, 47)])
def test_calibrated_tracking_matches_oracle(dev, h, w):
    """opt_pose_calib_sim3 (tracker.py:326-406) vs the float64 oracle; also constrain_points_to_ray.  64-wide rows take the
    four-points-per-lane path of k_track_accum_calib, widths that are not multiples of 4 (and odd point counts) the
    one-point path."""
    K = np.array([[float(w), 0, w / 2], [0, float(w), h / 2], [0, 0, 1]], dtype=np.float32)
    pr = synthetic.tracking_problem(h, w, seed=8)
    Xk_c = ot.constrain_points_to_ray((h, w), pr["Xk"].astype(np.float64), K.astype(np.float64))
    out = tracker.constrain_points_to_ray((h, w), _t(pr["Xk"], dev), K)
    assert np.abs(out.cpu().numpy() - Xk_c).max() < 1e-5
    Xf = pr["Xf_canon"][pr["idx"]]
    Tf, Trel, info = tracker.opt_pose_calib_sim3(_t(Xf, dev), _t(pr["Xk"], dev), _t(pr["T_WCf"], dev), _t(pr["T_WCk"], dev),
                                                 _t(pr["Qk"], dev), _t(pr["valid"], dev), K, (h, w))
    To, Trel_o, io = ot.opt_pose_calib_sim3(Xf, pr["Xk"], pr["T_WCf"], pr["T_WCk"], pr["Qk"], pr["valid"], K, (h, w))
    info = info.cpu().numpy()
    assert int(info[0]) == io["iters"]
    assert np.abs(Trel.cpu().numpy() - Trel_o).max() < 1e-4          # float32 per-point terms, pixel-scale residuals
    assert abs(info[1] - io["costs"][-1]) <= 2e-3 * io["costs"][-1]
    # batched == single
    st = lambda a: torch.stack([_t(a, dev)] * 2)
    Tb, Trb, ib = tracker.opt_pose_calib_sim3(st(Xf), st(pr["Xk"]), st(pr["T_WCf"]), st(pr["T_WCk"]), st(pr["Qk"]),
                                              st(pr["valid"]), K, (h, w))
    assert torch.equal(Trb[0], Trel) and torch.equal(Trb[1], Trel)


# ------------------------------------------------------------------ FrameTracker.track end to end (tracker.py:51-175)
def _track_scene(h, w, seed, dev):
    """A solvable frame / keyframe pair: synthetic.geometric_pair gives both views' points in the FRAME's coordinates and
    the true pixel correspondences; the keyframe's own canonical map is those points under a known Sim(3) (+ noise).
    Returns the host arrays and a match operator with mast3r_match_asymmetric's signature that injects them
    (tracker.track takes the operator as a callable, tracker.py:72-74)."""
    n = h * w
    sc = synthetic.geometric_pair(h, w, seed=seed, batch=1, noise=2e-4)
    rng = np.random.default_rng(seed + 100)
    uv = np.rint(sc["uv_true"][0]).astype(np.int64)
    inb = (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
    idx = np.clip(uv[:, 0], 0, w - 1) + w * np.clip(uv[:, 1], 0, h - 1)
    bad = rng.uniform(size=n) < 0.03                                   # 3 % wrong matches, flagged invalid
    idx[bad] = rng.integers(0, n, size=int(bad.sum()))
    vm = inb & ~bad & (rng.uniform(size=n) < 0.95)
    ang = np.deg2rad(1.5)
    q = np.array([0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)])
    T_true = np.concatenate([[0.03, -0.01, 0.02], q, [1.015]])
    Xkf = sc["X21"][0].reshape(n, 3).astype(np.float64)                # keyframe pixels' points in the frame's coordinates
    Xk_canon = (S.sim3_act_mlx(T_true, Xkf) + rng.normal(0, 2e-4, (n, 3))).astype(np.float32)
    u = lambda lo, hi: rng.uniform(lo, hi, size=(n, 1)).astype(np.float32)
    host = dict(idx=idx, vm=vm, Xff=sc["X11"][0].reshape(n, 3), Xkf=Xkf.astype(np.float32), Cff=u(0.5, 3.0), Ckf=u(0.5, 3.0),
                Qff=u(1.0, 4.0), Qkf=u(1.0, 4.0), Xk_canon=Xk_canon, Ck=u(0.5, 3.0), T_true=T_true)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def match_fn(model, frame, keyframe, idx_i2j_init=None):
        return (d(idx)[None], d(vm)[None, :, None], d(host["Xff"])[None], d(host["Cff"])[None], d(host["Qff"])[None],
                d(host["Xkf"])[None], d(host["Ckf"])[None], d(host["Qkf"])[None])
    return host, match_fn


def _oracle_track(z, T_WCf, T_WCk, cfg):
    """The same frame through the oracle pieces chained in the reference's order (tracker.py:88-158)."""
    from oracle import frame as of
    fr = of.FrameState(cfg["filtering_mode"])
    fr.update_pointmap(z["Xff"], z["Cff"])                                              # :94
    kf = of.FrameState(cfg["filtering_mode"])
    kf.update_pointmap(z["Xk_canon"], z["Ck"])
    idx, vm = z["idx"], z["vm"][:, None]
    Qk = ot.match_quality(z["Qff"][:, 0], z["Qkf"][:, 0], idx)[:, None]                 # :88-91
    Xf, Cf, Ck = fr.X_canon[idx], fr.get_average_conf()[idx], kf.get_average_conf()      # :177-214
    valid_opt, valid_kf = ot.validity(vm, Cf, Ck, Qk, cfg["C_conf"], cfg["Q_conf"])      # :108-113
    match_frac = float(valid_opt.astype(np.float32).sum()) / valid_opt.size
    T_f, T_rel, info = ot.opt_pose_ray_dist_sim3(Xf, kf.X_canon, T_WCf, T_WCk, Qk, valid_opt, cfg)
    kf.update_pointmap(ot.act_sim3(T_rel, z["Xkf"].astype(np.float64)), z["Ckf"])       # :146-147
    mk, uf = of.keyframe_stats(idx, vm, valid_kf)                                        # :149-155
    return dict(T_WCf=T_f, T_rel=T_rel, iters=info["iters"], kf=kf, fr=fr, match_frac=match_frac, match_frac_k=mk,
                unique_frac_f=uf, new_kf=min(mk, uf) < cfg["match_frac_thresh"])


@pytest.mark.parametrize("thresh", [0.333, 0.95])
def test_frame_tracker_track_matches_the_oracle_chain(dev, thresh):
    """FrameTracker.track (tracker.py:51-175) on solvable geometry with the true correspondences injected through the
    match operator: frame pointmap update, gathers and masks, the Gauss-Newton solve, the fused keyframe update
    (Sim3.act + weighted fusion in one kernel), the keyframe statistics and the new-keyframe decision, against the
    oracle pieces chained in the reference's order.  Two thresholds so that both outcomes of the decision occur."""
    from types import SimpleNamespace
    from mast3r_slam import config
    from mast3r_slam.frame import Keyframes, create_frame
    h, w = 96, 128
    n = h * w
    z, match_fn = _track_scene(h, w, 4, dev)
    T_WCk = np.array([0.2, -0.1, 0.05, 0.0, 0.0, np.sin(0.05), np.cos(0.05), 1.1], dtype=np.float32)
    config.set_config({"tracking": {"match_frac_thresh": thresh}})
    try:
        cfg = config.get_config()["tracking"]
        img = torch.zeros((3, h, w), dtype=torch.uint8, device=dev)
        kf = create_frame(0, img, T_WC=_t(T_WCk[None], dev))
        kf.update_pointmap(_t(z["Xk_canon"], dev), _t(z["Ck"], dev))
        frame = create_frame(1, img, T_WC=_t(T_WCk[None], dev))                          # slam.py: a new frame starts at the last pose
        kfs = Keyframes()
        kfs.append(kf)
        tr = tracker.FrameTracker(SimpleNamespace(device=dev), kfs)
        new_kf, match_info, try_reloc = tr.track(frame, mast3r_match_fn=match_fn)
        o = _oracle_track(z, T_WCk.astype(np.float64), T_WCk.astype(np.float64), cfg)
    finally:
        config.reset_config()
    assert try_reloc is False and len(match_info) == 6
    assert o["match_frac"] > 0.5                                                         # the gate was passed on real matches
    info = tr.last_info.cpu().numpy().reshape(-1)
    assert int(info[0]) == o["iters"] and info[3] == 1.0                                 # same stopping iteration, converged
    assert np.abs(frame.T_WC.cpu().numpy().reshape(8) - o["T_WCf"]).max() < 5e-5         # pose (t, q, s are O(1))
    rel_true = np.abs(o["T_rel"] - z["T_true"]).max()
    assert rel_true < 2e-3                                                               # ... and both found the scene's Sim(3)
    kfo = o["kf"]
    got = kfs.last_keyframe()
    assert got.N == kfo.N == 2 and frame.N == 1
    assert np.abs(got.X_canon.cpu().numpy() - kfo.X_canon).max() < 2e-5 * np.abs(kfo.X_canon).max()   # fused keyframe map
    assert np.allclose(got.C.cpu().numpy(), kfo.C, rtol=1e-6)
    assert np.array_equal(frame.X_canon.cpu().numpy(), o["fr"].X_canon.astype(np.float32))
    assert tr.last_stats == (o["match_frac_k"], o["unique_frac_f"])                      # integer counts / N: exact
    assert new_kf is o["new_kf"] and new_kf is (thresh > 0.5)
    assert (tr.idx_f2k is None) == new_kf                                                # :160-161


def test_frame_tracker_solve_failure_relocalises(dev):
    """tracker.py:121-141: a failing optimisation makes track() return (False, [], True) and leaves the frame pose and the
    keyframe map untouched.  Degenerate input: every matched point is the same point (rank-deficient normal matrix up
    to the 1e-6 regulariser -> the step's scale component leaves the float range), which the device solve flags as
    status 2."""
    from types import SimpleNamespace
    from mast3r_slam import config
    from mast3r_slam.frame import Keyframes, create_frame
    h, w = 32, 48
    n = h * w
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    one = np.tile(np.array([[0.3, -0.2, 2.0]], dtype=np.float32), (n, 1))
    far = np.tile(np.array([[30.0, 10.0, 2.0e4]], dtype=np.float32), (n, 1))             # a wildly inconsistent keyframe map
    conf = np.full((n, 1), 2.0, dtype=np.float32)

    def match_fn(model, frame, keyframe, idx_i2j_init=None):
        return (d(np.zeros(n, np.int64))[None], d(np.ones(n, bool))[None, :, None], d(one)[None], d(conf)[None], d(conf)[None],
                d(one)[None], d(conf)[None], d(conf)[None])
    img = torch.zeros((3, h, w), dtype=torch.uint8, device=dev)
    kf = create_frame(0, img)
    kf.update_pointmap(d(far), d(conf))
    frame = create_frame(1, img)
    kfs = Keyframes()
    kfs.append(kf)
    tr = tracker.FrameTracker(SimpleNamespace(device=dev), kfs)
    pose0, map0, n0 = frame.T_WC.clone(), kf.X_canon.clone(), kf.N
    out = tr.track(frame, mast3r_match_fn=match_fn)
    status = float(tr.last_info.reshape(-1)[3])
    assert status == 2.0, status
    assert out == (False, [], True)
    assert torch.equal(frame.T_WC, pose0) and torch.equal(kfs.last_keyframe().X_canon, map0) and kfs.last_keyframe().N == n0


def test_frame_tracker_with_the_fast_reciprocal_nn_matcher(dev):
    """matching.use_fast_nn behind the tracker: the match operator runs matching.match (-> match_fast_nn) on the scene's
    descriptor maps, so idx / valid are SPARSE (one match per seed at most).  The tracker's gates count fractions of the
    seeds then: the frame is not skipped, the Gauss-Newton solve on ~1.5 % of the pixels still finds the scene's Sim(3),
    and the keyframe statistics are fractions of the seed count."""
    from types import SimpleNamespace
    from mast3r_slam import config, matching
    from mast3r_slam.frame import Keyframes, create_frame
    h, w = 96, 128
    n = h * w
    z, _ = _track_scene(h, w, 9, dev)
    sc = synthetic.geometric_pair(h, w, seed=9, batch=1, noise=2e-4)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    X11, X21, D11, D21 = d(sc["X11"]), d(sc["X21"]), d(sc["D11"]), d(sc["D21"])
    seen = {}

    def match_fn(model, frame, keyframe, idx_i2j_init=None):
        idx, valid = matching.match(X11, X21, D11, D21, idx_i2j_init)
        seen["valid"] = int(valid.sum())
        return (idx, valid, d(z["Xff"])[None], d(z["Cff"])[None], d(z["Qff"])[None], d(z["Xkf"])[None], d(z["Ckf"])[None], d(z["Qkf"])[None])
    T_WCk = np.array([0.2, -0.1, 0.05, 0.0, 0.0, np.sin(0.05), np.cos(0.05), 1.1], dtype=np.float32)
    sub = 4
    config.set_config({"matching": {"use_fast_nn": True, "fast_nn_subsample": sub, "fast_nn_rounds": 4}, "tracking": {"match_frac_thresh": 0.3}})
    try:
        img = torch.zeros((3, h, w), dtype=torch.uint8, device=dev)
        kf = create_frame(0, img, T_WC=_t(T_WCk[None], dev))
        kf.update_pointmap(_t(z["Xk_canon"], dev), _t(z["Ck"], dev))
        frame = create_frame(1, img, T_WC=_t(T_WCk[None], dev))
        kfs = Keyframes()
        kfs.append(kf)
        tr = tracker.FrameTracker(SimpleNamespace(device=dev), kfs)
        new_kf, match_info, try_reloc = tr.track(frame, mast3r_match_fn=match_fn)
    finally:
        config.reset_config()
    seeds = (h // sub) * (w // sub)
    assert 0.5 * seeds < seen["valid"] <= seeds and seen["valid"] < 0.1 * n            # sparse: far below min_match_frac of the PIXELS
    assert try_reloc is False and len(match_info) == 6
    info = tr.last_info.cpu().numpy().reshape(-1)
    assert info[3] == 1.0                                                              # converged
    T_rel = S.sim3_mul_mlx(S.sim3_inv_mlx(T_WCk.astype(np.float64)), frame.T_WC.cpu().numpy().reshape(8).astype(np.float64))
    assert np.abs(T_rel - z["T_true"]).max() < 3e-3                                    # the scene's Sim(3) from the sparse matches
    mk, uf = tr.last_stats
    assert 0.4 < mk <= 1.0 and 0.4 < uf <= 1.0 and new_kf is False                     # fractions of the seeds
