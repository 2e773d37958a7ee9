"""GPU parity of the frame / keyframe state kernels (csrc/frame.hip, through the C ABI) against the numpy
oracle (oracle/frame.py): every filtering mode of Frame.update_pointmap, the fused Sim3.act variant of the
tracker's keyframe update, and the unique-match count.  float32 on both sides: tolerances 1e-6 (the fused
multiply-adds of the device) and 2e-5 for the spherical mode (sin/cos/atan2/acos of two libraries)."""
import numpy as np
import pytest
import torch

from mast3r_slam import config, tracker
from mast3r_slam.frame import create_frame
from oracle import frame as OF
from oracle import sim3 as OS

pytestmark = pytest.mark.gpu


def _cloud(n, seed):
    r = np.random.default_rng(seed)
    X = r.normal(size=(n, 3)).astype(np.float32) + np.array([0, 0, 3], np.float32)
    C = r.uniform(0.5, 4.0, size=(n, 1)).astype(np.float32)
    return X, C


@pytest.mark.parametrize("mode", ["first", "recent", "best_score", "indep_conf", "weighted_pointmap", "weighted_spherical"])
def test_update_pointmap_modes(dev, mode):
    n = 70001                                                                  # ragged: not a multiple of the block
    config.set_config({"tracking": {"filtering_mode": mode, "filtering_score": "median"}})
    try:
        f = create_frame(0, torch.zeros(3, 16, 16, device=dev))
        o = OF.FrameState(mode, "median")
        for k, scale in enumerate((1.0, 0.6, 1.7, 1.2)):
            X, C = _cloud(n, 10 + k)
            C = (C * scale).astype(np.float32)
            f.update_pointmap(torch.from_numpy(X).to(dev), torch.from_numpy(C).to(dev))
            o.update_pointmap(X, C)
            tol = 2e-5 if mode == "weighted_spherical" else 1e-6
            assert np.abs(f.X_canon.cpu().numpy() - o.X_canon).max() < tol * 8
            assert np.allclose(f.C.cpu().numpy(), o.C, rtol=1e-6, atol=0)
            assert (f.N, f.N_updates) == (o.N, o.N_updates)
        assert np.allclose(f.get_average_conf().cpu().numpy(), o.get_average_conf(), rtol=1e-6)
    finally:
        config.set_config({"tracking": {"filtering_mode": "weighted_pointmap"}})


@pytest.mark.parametrize("n", [1, 2, 5, 6, 1000, 70001, 262144])
def test_device_median_is_exact(dev, n):
    """m3_median_f32 (the score of "best_score" filtering, frame.py:59-73): exact radix select on the float bits - equal
    to np.median bit for bit for odd and even counts, heavy duplicates, negative values and signed zeros."""
    from mast3r_slam import _ffi
    L = _ffi.lib()
    rng = np.random.default_rng(n)
    cases = [rng.normal(size=n).astype(np.float32),
             (rng.integers(0, 7, size=n) / 3.0).astype(np.float32),                     # few distinct values
             np.concatenate([np.zeros(n // 2, np.float32), -np.zeros(n - n // 2, np.float32)]),
             (rng.uniform(0.5, 4.0, size=n) * 10.0 ** rng.integers(-20, 20, size=n)).astype(np.float32)]
    for v in cases:
        d = torch.from_numpy(v).to(dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.m3_median_ws_words()), dtype=torch.int32, device=dev)
        _ffi.call("m3_median_f32", _ffi.ptr(d), n, _ffi.ptr(ws), _ffi.ptr(out), _ffi.stream_ptr())
        ref = np.median(v)
        got = out.cpu().numpy()[0]
        assert got == ref or (np.isnan(got) and np.isnan(ref)), (n, got, ref)


@pytest.mark.parametrize("score", ["median", "mean"])
def test_best_score_mode_decides_on_the_device(dev, score):
    """Winner takes all (frame.py:103-107) with an EVEN point count and both score kinds; the decision and the running best
    score live in a device buffer (no host synchronisation inside update_pointmap)."""
    n = 4096
    config.set_config({"tracking": {"filtering_mode": "best_score", "filtering_score": score}})
    try:
        f = create_frame(0, torch.zeros(3, 16, 16, device=dev))
        o = OF.FrameState("best_score", score)
        for k, scale in enumerate((1.0, 0.6, 1.7, 1.2, 1.7, 2.5)):
            X, C = _cloud(n, 40 + k)
            C = (C * scale).astype(np.float32)
            f.update_pointmap(torch.from_numpy(X).to(dev), torch.from_numpy(C).to(dev))
            o.update_pointmap(X, C)
            assert np.array_equal(f.X_canon.cpu().numpy(), o.X_canon) and np.array_equal(f.C.cpu().numpy(), o.C)
            assert (f.N, f.N_updates) == (o.N, o.N_updates)
            assert abs(f._score - o._score) <= 1e-6 * abs(o._score)
    finally:
        config.set_config({"tracking": {"filtering_mode": "weighted_pointmap", "filtering_score": "median"}})


def test_best_score_after_weighted_updates_keeps_n_when_the_new_score_loses(dev):
    """frame.py:103-109 resets N only when the new score WINS (round-3 advisor finding: the device path set N = 1
    unconditionally).  Two weighted_pointmap updates (N = 2), then the frame's mode is overridden to best_score: a losing
    update must leave N, the pointmap and get_average_conf() = C / N alone, a winning one resets N to 1."""
    n = 2048
    f = create_frame(0, torch.zeros(3, 16, 16, device=dev))
    o = OF.FrameState("weighted_pointmap")
    for k in range(2):
        X, C = _cloud(n, 70 + k)
        f.update_pointmap(torch.from_numpy(X).to(dev), torch.from_numpy(C).to(dev))
        o.update_pointmap(X, C)
    assert f.N == o.N == 2
    f.filtering_mode = o.mode = "best_score"
    o._score = f._score = 1e9                                           # the stored best beats anything below
    X, C = _cloud(n, 80)
    f.update_pointmap(torch.from_numpy(X).to(dev), torch.from_numpy(C).to(dev))
    o.update_pointmap(X, C)
    # (the weighted fusion itself is fp32 on both sides but contracts differently: last-bit differences, as in the mode test)
    assert f.N == o.N == 2 and np.abs(f.X_canon.cpu().numpy() - o.X_canon).max() < 5e-6
    assert np.allclose(f.get_average_conf().cpu().numpy(), o.get_average_conf(), rtol=1e-6)
    o._score = f._score = 0.0                                           # ... and now anything wins
    f.update_pointmap(torch.from_numpy(X).to(dev), torch.from_numpy(C).to(dev))
    o.update_pointmap(X, C)
    assert f.N == o.N == 1 and np.array_equal(f.X_canon.cpu().numpy(), o.X_canon)
    assert np.array_equal(f.get_average_conf().cpu().numpy(), o.get_average_conf())


def test_fused_sim3_act_update(dev):
    """keyframe.update_pointmap(T_CkCf.act(Xkf), Ckf) (tracker.py:146-147) in one kernel."""
    n = 4096
    X, C = _cloud(n, 1)
    Y, D = _cloud(n, 2)
    T = np.array([[0.1, -0.2, 0.05, 0.02, -0.03, 0.04, 0.0, 1.07]], np.float32)
    T[0, 6] = np.sqrt(1 - (T[0, 3:6] ** 2).sum())
    f = create_frame(0, torch.zeros(3, 16, 16, device=dev))
    f.update_pointmap(torch.from_numpy(X).to(dev), torch.from_numpy(C).to(dev))
    f.update_pointmap(torch.from_numpy(Y).to(dev), torch.from_numpy(D).to(dev), T=torch.from_numpy(T).to(dev))
    o = OF.FrameState("weighted_pointmap")
    o.update_pointmap(X, C)
    o.update_pointmap(OS.sim3_act_mlx(T, Y).astype(np.float32), D)
    assert np.abs(f.X_canon.cpu().numpy() - o.X_canon).max() < 5e-6
    src = torch.from_numpy(X).to(dev)
    g = create_frame(1, torch.zeros(3, 16, 16, device=dev))
    g.update_pointmap(src, torch.from_numpy(C).to(dev))
    g.update_pointmap(torch.from_numpy(Y).to(dev), torch.from_numpy(D).to(dev))
    assert torch.equal(src.cpu(), torch.from_numpy(X))                        # the frame fuses into its own buffers


def test_count_unique(dev):
    r = np.random.default_rng(0)
    n = 262144
    idx = r.integers(0, n, size=n)
    idx[: n // 3] = idx[n // 3: 2 * (n // 3)][: n // 3]                        # plenty of duplicates
    valid = r.random(n) < 0.7
    got = int(tracker.count_unique(torch.from_numpy(idx).to(dev), torch.from_numpy(valid).to(dev), n).cpu())
    assert got == np.unique(idx[valid]).shape[0]
    assert int(tracker.count_unique(torch.from_numpy(idx).to(dev), torch.zeros(n, dtype=torch.bool, device=dev), n).cpu()) == 0
    with pytest.raises(RuntimeError):
        tracker.count_unique(torch.from_numpy(idx), torch.from_numpy(valid), n)   # CPU tensors are rejected
