"""CPU known-answer tests of the frame-state oracle (oracle/frame.py).  The reference module needs mlx,
so these pin the restatement analytically ('parity unpinned' for the MLX-only wrappers)."""
import numpy as np
import pytest

from oracle import frame as OF


def _cloud(n, seed):
    r = np.random.default_rng(seed)
    X = r.normal(size=(n, 3)).astype(np.float32) + np.array([0, 0, 3], np.float32)
    C = r.uniform(0.5, 4.0, size=(n, 1)).astype(np.float32)
    return X, C


def test_spherical_round_trip_and_axes():
    X, _ = _cloud(500, 0)
    S = OF.cartesian_to_spherical(X)
    assert np.allclose(OF.spherical_to_cartesian(S), X, atol=2e-6)
    s = OF.cartesian_to_spherical(np.array([[0, 0, 2.0], [1.0, 0, 0], [0, 3.0, 0]], np.float32))
    assert np.allclose(s[:, 0], [2, 1, 3], atol=1e-6)                       # r
    assert np.allclose(s[:, 2], [0, np.pi / 2, np.pi / 2], atol=1e-6)       # theta from +z
    assert np.allclose(s[1:, 1], [0, np.pi / 2], atol=1e-6)                 # phi from +x


@pytest.mark.parametrize("mode", ["weighted_pointmap", "weighted_spherical"])
def test_weighted_modes_are_confidence_weighted_means(mode):
    X, C = _cloud(300, 1)
    f = OF.FrameState(mode)
    f.update_pointmap(X, C)
    f.update_pointmap(X, 3 * C)                                             # same points again: unchanged, C sums
    assert np.allclose(f.X_canon, X, atol=2e-5) and np.allclose(f.C, 4 * C) and f.N == 2 and f.N_updates == 2
    assert np.allclose(f.get_average_conf(), 2 * C)
    if mode == "weighted_pointmap":
        Y, D = _cloud(300, 2)
        f = OF.FrameState(mode); f.update_pointmap(X, C); f.update_pointmap(Y, D)
        assert np.allclose(f.X_canon, (C * X + D * Y) / (C + D), atol=1e-6)


def test_replace_modes_and_indep_conf():
    X, C = _cloud(200, 3)
    Y, D = _cloud(200, 4)
    Z, E = _cloud(200, 5)
    f = OF.FrameState("first"); [f.update_pointmap(*p) for p in ((X, C), (Y, D), (Z, E))]
    assert np.array_equal(f.X_canon, Y) and f.N == 1 and f.N_updates == 3     # frame.py:93-97: the 2nd update wins, later ones ignored
    f = OF.FrameState("recent"); [f.update_pointmap(*p) for p in ((X, C), (Y, D), (Z, E))]
    assert np.array_equal(f.X_canon, Z) and np.array_equal(f.C, E)
    f = OF.FrameState("indep_conf"); f.update_pointmap(X, C); f.update_pointmap(Y, D)
    take = (D > C)
    assert np.array_equal(f.X_canon, np.where(take, Y, X)) and np.array_equal(f.C, np.maximum(C, D)) and 0 < take.mean() < 1
    f = OF.FrameState("best_score", "mean"); f.update_pointmap(X, C); f.update_pointmap(Y, 0.5 * C); f.update_pointmap(Z, 2 * C)
    assert np.array_equal(f.X_canon, Z) and f.N_updates == 3                  # lower score ignored, higher score replaces


def test_keyframe_stats():
    idx = np.array([[0, 1, 1, 2, 2, 2, 7, 7]])
    vm = np.array([[1, 1, 1, 1, 0, 1, 0, 0]], bool)
    vk = np.array([1, 1, 0, 0, 1, 1, 0, 0], bool)
    mf, uf = OF.keyframe_stats(idx, vm, vk)
    assert mf == 0.5 and uf == 3 / 8                                          # distinct valid values {0,1,2}
