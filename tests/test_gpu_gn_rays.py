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
