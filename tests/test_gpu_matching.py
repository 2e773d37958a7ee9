"""GPU parity: HIP matching kernels (through the C ABI) vs the CPU oracle and the golden
vectors.  Integer/index outputs and float32 positions are compared BIT-EXACT."""
import os

import numpy as np
import pytest
import torch

from mast3r_slam import config, kernels, matching, synthetic
from oracle import matching as om

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("with_idx", [False, True])
def test_prep_bit_exact(dev, with_idx):
    sc = synthetic.geometric_pair(48, 64, seed=1, batch=2)
    idx = None
    if with_idx:
        idx = np.random.default_rng(0).integers(0, 48 * 64, size=(2, 48 * 64)).astype(np.int64)
    ro, to, po = om.prep_for_iter_proj(sc["X11"], sc["X21"], idx)
    r, t, p = matching.prep_for_iter_proj(_t(sc["X11"], dev), _t(sc["X21"], dev), None if idx is None else _t(idx, dev))
    assert np.array_equal(r.cpu().numpy(), ro)
    assert np.array_equal(t.cpu().numpy(), to)
    assert np.array_equal(p.cpu().numpy(), po)


@pytest.mark.parametrize("tag", ["b1", "b2", "earlystop"])
def test_iter_proj_golden_bit_exact(dev, golden_dir, tag):
    z = _load(golden_dir, f"iter_proj_{tag}.npz")
    p, v = kernels.iter_proj(_t(z["rays_with_grad"], dev), _t(z["pts3d_norm"], dev), _t(z["p_init"], dev),
                             int(z["max_iter"]), float(z["lambda_init"]), float(z["convergence_thresh"]))
    assert np.array_equal(p.cpu().numpy(), z["p_ref"])          # vs the REFERENCE's numpy twin
    assert np.array_equal(v.cpu().numpy(), z["valid_ref"])


@pytest.mark.parametrize("tag", ["edge_it1", "edge_it3", "edge_lam", "refbench"])
def test_iter_proj_edge_case_goldens_bit_exact(dev, golden_dir, tag):
    """Outputs of the REFERENCE's numpy twin on a random ray map: steps that leave the image, the determinant clamp, starts
    outside the image and on its corners, N != H * W (the untiled thread mapping), 1 / 3 / 10 iterations, lambda 1e-2."""
    z = _load(golden_dir, f"iter_proj_{tag}.npz")
    p, v = kernels.iter_proj(_t(z["rays_with_grad"], dev), _t(z["pts3d_norm"], dev), _t(z["p_init"], dev),
                             int(z["max_iter"]), float(z["lambda_init"]), float(z["convergence_thresh"]))
    assert np.array_equal(p.cpu().numpy(), z["p_ref"])
    assert np.array_equal(v.cpu().numpy(), z["valid_ref"])


@pytest.mark.parametrize("tag", ["r2_d16", "r4_d24", "r1_d64", "r3_d5"])
def test_refine_other_lengths_and_radii_goldens_bit_exact(dev, golden_dir, tag):
    """Outputs of the REFERENCE's numpy twin for D = 16 / 24 / 64 / 5 and radius 2 / 4 / 1 / 3: the vector kernels of the
    other descriptor lengths and the generic one, border handling, ties."""
    z = _load(golden_dir, f"refine_matches_{tag}.npz")
    r = kernels.refine_matches(_t(z["D11"], dev), _t(z["D21"], dev), _t(z["p1"], dev), int(z["radius"]), int(z["dilation_max"]))
    assert np.array_equal(r.cpu().numpy(), z["p_ref"])


def test_iter_proj_numpy_in_numpy_out_like_reference(golden_dir):
    z = _load(golden_dir, "iter_proj_b1.npz")
    p, v = kernels.iter_proj(z["rays_with_grad"], z["pts3d_norm"], z["p_init"], 10, 1e-8, 1e-6)
    assert isinstance(p, np.ndarray) and p.dtype == np.float32 and v.dtype == np.bool_
    assert np.array_equal(p, z["p_ref"]) and np.array_equal(v, z["valid_ref"])


@pytest.mark.parametrize("scope", ["global", "batch"])
def test_iter_proj_scopes_and_early_stop(dev, scope):
    # two batch items that converge at different iterations
    h, w = 24, 32
    vv, uu = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    X11, X21 = [], []
    for amp in (0.02, 2.5):
        bump = np.sin(np.pi * uu / (w - 1)) * np.sin(np.pi * vv / (h - 1))
        X11.append(synthetic._surface(uu, vv, h, w))
        X21.append(synthetic._surface(uu + amp * bump, vv - 0.5 * amp * bump, h, w))
    X11 = np.stack(X11).astype(np.float32); X21 = np.stack(X21).astype(np.float32)
    rays, tgt, p0 = om.prep_for_iter_proj(X11, X21, None)
    for thr in (0.05, 1e-6):
        po, vo = om.iter_proj(rays, tgt, p0, 10, 1e-8, thr, scope)
        p, v = kernels.iter_proj(_t(rays, dev), _t(tgt, dev), _t(p0, dev), 10, 1e-8, thr, stop_scope=scope)
        assert np.array_equal(p.cpu().numpy(), po), (scope, thr)
        assert np.array_equal(v.cpu().numpy(), vo)
    # the two scopes really differ on this input
    pg, _ = om.iter_proj(rays, tgt, p0, 10, 1e-8, 0.05, "global")
    pb, _ = om.iter_proj(rays, tgt, p0, 10, 1e-8, 0.05, "batch")
    assert not np.array_equal(pg, pb)


def test_iter_proj_white_noise_ragged_and_edge_sizes(dev):
    rng = np.random.default_rng(42)
    # reference benchmark recipe (benchmark_all_kernels.py:62-76): random rays, ragged N != H*W
    for (b, h, w, n) in ((1, 64, 64, 1000), (2, 37, 53, 777), (1, 2, 2, 5)):
        rays = rng.normal(size=(b, h, w, 9)).astype(np.float32)
        pts = rng.normal(size=(b, n, 3)).astype(np.float32)
        pts /= np.linalg.norm(pts, axis=-1, keepdims=True)
        p0 = (rng.uniform(size=(b, n, 2)) * [w - 1, h - 1]).astype(np.float32)
        for it in (0, 1, 10):
            po, vo = om.iter_proj(rays, pts, p0, it, 1e-8, 1e-6, "global")
            p, v = kernels.iter_proj(_t(rays, dev), _t(pts, dev), _t(p0, dev), it, 1e-8, 1e-6)
            assert np.array_equal(p.cpu().numpy(), po, equal_nan=True), (b, h, w, n, it)
            assert np.array_equal(v.cpu().numpy(), vo)
    # empty input
    p, v = kernels.iter_proj(torch.zeros(1, 4, 4, 9, device=dev), torch.zeros(1, 0, 3, device=dev),
                             torch.zeros(1, 0, 2, device=dev))
    assert p.shape == (1, 0, 2) and v.shape == (1, 0)


@pytest.mark.parametrize("dmax", [0, 2])
def test_refine_golden_bit_exact(dev, golden_dir, dmax):
    z = _load(golden_dir, f"refine_matches_d{dmax}.npz")
    r = kernels.refine_matches(_t(z["D11"], dev), _t(z["D21"], dev), _t(z["p1"], dev), int(z["radius"]),
                               int(z["dilation_max"]))
    assert r.dtype == torch.int32
    assert np.array_equal(r.cpu().numpy(), z["p_ref"])          # vs the REFERENCE's numpy twin


@pytest.mark.parametrize("d", [24, 16, 32, 64, 20, 3])
@pytest.mark.parametrize("chained", [False, True])
def test_refine_vs_oracle_all_descriptor_sizes(dev, d, chained):
    rng = np.random.default_rng(d)
    b, h, w, n = 2, 30, 41, 900
    D11 = rng.normal(size=(b, h, w, d)).astype(np.float32)
    D21 = rng.normal(size=(b, n, d)).astype(np.float32)
    p1 = np.stack([rng.integers(-4, w + 4, size=(b, n)), rng.integers(-4, h + 4, size=(b, n))], -1).astype(np.int32)
    ro = om.refine_matches(D11, D21, p1, 3, 2, chained=chained)
    r = kernels.refine_matches(_t(D11, dev), _t(D21, dev), _t(p1, dev), 3, 2, chained=chained)
    assert np.array_equal(r.cpu().numpy(), ro)
    # ties: constant descriptors -> first candidate in raster order must win
    D11c = np.ones_like(D11); D21c = np.ones_like(D21)
    rc = kernels.refine_matches(_t(D11c, dev), _t(D21c, dev), _t(p1, dev), 2, 0).cpu().numpy()
    assert np.array_equal(rc, om.refine_matches(D11c, D21c, p1, 2, 0))


def test_match_iterative_proj_end_to_end_bit_exact(dev):
    sc = synthetic.geometric_pair(48, 64, seed=21, batch=2)
    config.set_config({"matching": {"use_simple": False}})
    try:
        for idx_init in (None, np.random.default_rng(1).integers(0, 48 * 64, size=(2, 48 * 64)).astype(np.int64)):
            io, vo = om.match_iterative_proj(sc["X11"], sc["X21"], sc["D11"], sc["D21"], idx_init, dilation_max=2)
            i, v = matching.match(_t(sc["X11"], dev), _t(sc["X21"], dev), _t(sc["D11"], dev), _t(sc["D21"], dev),
                                  None if idx_init is None else _t(idx_init, dev))
            assert i.dtype == torch.int64 and v.dtype == torch.bool and v.shape == (2, 48 * 64, 1)
            assert np.array_equal(i.cpu().numpy(), io)
            assert np.array_equal(v.cpu().numpy(), vo)
    finally:
        config.reset_config()


def test_match_simple_bit_exact(dev):
    sc = synthetic.geometric_pair(32, 48, seed=4, batch=2)
    idx = np.random.default_rng(2).integers(0, 32 * 48, size=(2, 32 * 48)).astype(np.int64)
    for ii in (None, idx):
        io, vo = om.match_simple(sc["X11"], sc["X21"], ii)
        i, v = matching.match(_t(sc["X11"], dev), _t(sc["X21"], dev), None, None, None if ii is None else _t(ii, dev))
        assert np.array_equal(i.cpu().numpy(), io) and np.array_equal(v.cpu().numpy(), vo)


def test_full_size_512_properties(dev):
    """BASELINE size: properties that do not need the (slow) oracle at 262 144 points."""
    h = w = 512
    sc = synthetic.geometric_pair(h, w, seed=0, batch=2)
    config.set_config({"matching": {"use_simple": False}})
    try:
        args = [_t(sc[k], dev) for k in ("X11", "X21", "D11", "D21")]
        i1, v1 = matching.match(*args)
        i2, v2 = matching.match(*args)
        assert torch.equal(i1, i2) and torch.equal(v1, v2)                  # deterministic
        for b in range(2):                                                    # batched == loop of singles
            ib, vb = matching.match(*[a[b:b + 1] for a in args])
            assert torch.equal(ib[0], i1[b]) and torch.equal(vb[0], v1[b])
        idx = i1.cpu().numpy(); val = v1.cpu().numpy()[..., 0]
        assert val.mean() > 0.9
        assert (idx[val] >= 0).all() and (idx[val] < h * w).all()
        u, v = idx % w, idx // w
        err = np.hypot(u - sc["uv_true"][..., 0], v - sc["uv_true"][..., 1])
        assert np.median(err[val]) < 1.0 and (err[val] < 2.5).mean() > 0.99  # lands on the true match
        # oracle spot check on a crop-free subset: the first 4096 points through the L1 ops
        rays, tgt, p0 = matching.prep_for_iter_proj(args[0][:1], args[1][:1])
        p, vv = kernels.iter_proj(rays, tgt[:, :4096].contiguous(), p0[:, :4096].contiguous())
        po, vo = om.iter_proj(rays.cpu().numpy(), tgt[:, :4096].cpu().numpy(), p0[:, :4096].cpu().numpy())
        assert np.array_equal(p.cpu().numpy(), po) and np.array_equal(vv.cpu().numpy(), vo)
    finally:
        config.reset_config()


@pytest.mark.parametrize("features", ["fp32", "fp16"])
def test_full_size_512_match_bit_exact_vs_oracle(dev, features):
    """The size and the kernels that are TIMED (bench.py: 512x512, all 262 144 points, the LDS-tiled k_refine_lds<24> path
    with its halo staging and image borders, fp32 and half descriptor storage): index and validity of
    matching.match equal oracle.matching.match_iterative_proj (kernels.py:151-254, :496-537; matching.py:339-461) bit
    for bit on every point.  Two maps per call so that the second map's tiles start behind the first one's."""
    h = w = 512
    sc = synthetic.geometric_pair(h, w, seed=3, batch=2)
    D11, D21 = sc["D11"], sc["D21"]
    if features == "fp16":
        D11, D21 = D11.astype(np.float16), D21.astype(np.float16)
    config.set_config({"matching": {"use_simple": False}})
    try:
        i, v = matching.match(_t(sc["X11"], dev), _t(sc["X21"], dev), _t(D11, dev), _t(D21, dev))
        io, vo = om.match_iterative_proj(sc["X11"], sc["X21"], D11.astype(np.float32), D21.astype(np.float32), None,
                                         dilation_max=2)
    finally:
        config.reset_config()
    assert vo.mean() > 0.9                                            # the comparison is on matched points, not rejects
    assert np.array_equal(i.cpu().numpy(), io)
    assert np.array_equal(v.cpu().numpy(), vo)


# ----------------------------------------------------------------- "fp16 features" (BASELINE configs[4])
@pytest.mark.parametrize("d", [24, 16, 32, 64, 20])
@pytest.mark.parametrize("chained", [False, True])
def test_refine_fp16_descriptors_bit_exact_vs_oracle_on_rounded_values(dev, d, chained):
    """Half-stored descriptors are widened exactly and scored in fp32: the result must equal the oracle (and the
    fp32 kernel) run on the half-rounded values, bit for bit - the storage type changes bytes moved, not arithmetic."""
    rng = np.random.default_rng(100 + d)
    b, h, w, n = 2, 30, 41, 900
    D11 = rng.normal(size=(b, h, w, d)).astype(np.float16)
    D21 = rng.normal(size=(b, n, d)).astype(np.float16)
    p1 = np.stack([rng.integers(-4, w + 4, size=(b, n)), rng.integers(-4, h + 4, size=(b, n))], -1).astype(np.int32)
    ro = om.refine_matches(D11.astype(np.float32), D21.astype(np.float32), p1, 3, 2, chained=chained)
    r16 = kernels.refine_matches(_t(D11, dev), _t(D21, dev), _t(p1, dev), 3, 2, chained=chained)
    r32 = kernels.refine_matches(_t(D11, dev).float(), _t(D21, dev).float(), _t(p1, dev), 3, 2, chained=chained)
    assert r16.dtype == torch.int32
    assert np.array_equal(r16.cpu().numpy(), ro)
    assert torch.equal(r16, r32)
    with pytest.raises(ValueError, match="float16"):
        kernels.refine_matches(_t(D11, dev), _t(D21, dev).float(), _t(p1, dev), 3, 2)


def test_fp16_features_dense_match_lds_path_and_agreement(dev):
    """Tiled LDS-staged path (N == H*W, D == 24) with half descriptors: bit-exact against the oracle on the rounded
    descriptors at a size the oracle finishes in seconds; at 512x512 the half-feature matcher must agree with the
    fp32-feature matcher on all but near-tie points and land on the true match equally well."""
    config.set_config({"matching": {"use_simple": False}})
    try:
        sc = synthetic.geometric_pair(48, 64, seed=5, batch=2)
        D11h, D21h = sc["D11"].astype(np.float16), sc["D21"].astype(np.float16)
        io, vo = om.match_iterative_proj(sc["X11"], sc["X21"], D11h.astype(np.float32), D21h.astype(np.float32), None,
                                         dilation_max=2)
        i, v = matching.match(_t(sc["X11"], dev), _t(sc["X21"], dev), _t(D11h, dev), _t(D21h, dev))
        assert np.array_equal(i.cpu().numpy(), io) and np.array_equal(v.cpu().numpy(), vo)
        h = w = 512
        sc = synthetic.geometric_pair(h, w, seed=0, batch=2)
        X11, X21, D11, D21 = [_t(sc[k], dev) for k in ("X11", "X21", "D11", "D21")]
        i32, v32 = matching.match(X11, X21, D11, D21)
        i16, v16 = matching.match(X11, X21, D11.half(), D21.half())
        both = (v32 & v16)[..., 0]
        # the synthetic descriptors vary smoothly: neighbouring candidates score within the half rounding of each other,
        # so a fraction of the arg-maxes moves - by one pixel, never further
        assert float((i32 == i16)[both].float().mean()) > 0.85
        du, dv = (i32 % w - i16 % w).abs(), (i32 // w - i16 // w).abs()
        assert float((torch.maximum(du, dv)[both] <= 1).float().mean()) > 0.999
        idx = i16.cpu().numpy(); val = v16.cpu().numpy()[..., 0]
        err = np.hypot(idx % w - sc["uv_true"][..., 0], idx // w - sc["uv_true"][..., 1])
        assert val.mean() > 0.9 and np.median(err[val]) < 1.0 and (err[val] < 2.5).mean() > 0.99
    finally:
        config.reset_config()
