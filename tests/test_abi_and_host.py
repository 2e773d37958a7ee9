"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute
calls without a GPU), the host mirror keeps the reference's names, and the product path
fails loudly instead of falling back."""
import ctypes
import inspect
import os

import numpy as np
import pytest
import torch

from mast3r_slam import _ffi, config, kernels, matching, tracker


@pytest.fixture(scope="module")
def built_lib():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(_ffi.LIB_PATH):
        spec = importlib.util.spec_from_file_location("m3build", os.path.join(root, "mast3r-slam_amd", "build.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        m.build(verbose=False)
    return ctypes.CDLL(_ffi.LIB_PATH)


def test_every_declared_symbol_is_exported(built_lib):
    names = _ffi.declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(built_lib, n), f"{n} declared in include/ but not exported"


def test_metadata_entry_points(built_lib):
    L = _ffi.lib()
    assert L.m3_abi_version() == 2007          # exact: a signature change must bump it (include/m3slam.h)
    assert L.m3_chol_ws_doubles(1785) == 1 + 1785 + 28 * (2 * 64 * 64 + 1) and L.m3_chol_ws_doubles(7) == 1 + 7 + 2 * 4096 + 1
    assert L.m3_status_string(0) == b"ok"
    assert b"invalid" in L.m3_status_string(-1)
    assert L.m3_track_ws_doubles() > 36
    assert L.m3_gn_rays_chunks(262144) == 128 and L.m3_gn_rays_chunks(10) == 1
    assert L.m3_gn_rays_max_dim() % 7 == 0


def test_null_arguments_are_rejected_without_a_gpu(built_lib):
    # argument validation happens before any HIP call, so this is safe on a CPU-only box
    L = _ffi.lib()
    assert L.m3_iter_proj(None, None, None, None, None, None, 1, 4, 4, 16, 10, 1e-8, 1e-6, 0, None) == -1
    assert L.m3_refine_matches(None, None, None, None, 1, 4, 4, 24, 16, 3, 2, 0, None) == -1
    assert L.m3_track_gn_ray_dist(None, None, None, None, None, None, None, None, None, None, 16, 10, 1.345,
                                  0.003, 10.0, 1e-3, 1e-3, 0, None) == -1
    with pytest.raises(RuntimeError, match="invalid argument"):
        _ffi.call("m3_prep_iter_proj", None, None, None, None, None, None, 1, 4, 4, None)


def test_reference_operator_names_and_signatures():
    # kernels.py:107-115, :463-470, :262-279 of the reference
    assert list(inspect.signature(kernels.iter_proj).parameters)[:7] == [
        "rays_with_grad", "pts3d_norm", "p_init", "max_iter", "lambda_init", "convergence_thresh", "use_metal"]
    assert list(inspect.signature(kernels.refine_matches).parameters)[:6] == [
        "D11", "D21", "p1", "radius", "dilation_max", "use_metal"]
    assert list(inspect.signature(kernels.gauss_newton_rays).parameters)[:16] == [
        "Twc", "Xs", "Cs", "ii", "jj", "idx_ii2jj", "valid_match", "Q", "sigma_ray", "sigma_dist", "C_thresh",
        "Q_thresh", "max_iter", "delta_thresh", "pin", "use_metal"]
    for fn in ("match", "match_simple", "match_iterative_proj", "prep_for_iter_proj", "pixel_to_lin", "lin_to_pixel"):
        assert callable(getattr(matching, fn))
    assert list(inspect.signature(matching.match).parameters) == ["X11", "X21", "D11", "D21", "idx_1_to_2_init"]
    assert hasattr(tracker.FrameTracker, "track") and hasattr(tracker.FrameTracker, "reset_idx_f2k")
    # load_mast3r: the reference's parameter names AND defaults (mast3r_utils.py:47-52); the DUNE default is out of
    # scope and must say so instead of silently handing back another model
    from mast3r_slam import mast3r_utils
    sig = inspect.signature(mast3r_utils.load_mast3r).parameters
    assert [(k, sig[k].default) for k in list(sig)[:4]] == [
        ("model_type", "dunemast3r"), ("variant", "base"), ("resolution", 336), ("precision", "fp16")]
    with pytest.raises(NotImplementedError, match="mast3r_full"):
        mast3r_utils.load_mast3r()
    with pytest.raises(ValueError, match="Unknown model type"):
        mast3r_utils.load_mast3r("mast3r_small")
    # kernels.py:325-346, :396-412 (calibrated / point variants keep the rays signature's leading arguments)
    assert list(inspect.signature(kernels.gauss_newton_calib).parameters)[:20] == [
        "Twc", "Xs", "Cs", "K", "ii", "jj", "idx_ii2jj", "valid_match", "Q", "img_size", "pixel_border", "z_eps",
        "sigma_pixel", "sigma_depth", "C_thresh", "Q_thresh", "max_iter", "delta_thresh", "pin", "use_metal"]
    assert list(inspect.signature(kernels.gauss_newton_points).parameters)[:15] == [
        "Twc", "Xs", "Cs", "ii", "jj", "idx_ii2jj", "valid_match", "Q", "sigma_point", "C_thresh", "Q_thresh", "max_iter",
        "delta_thresh", "pin", "use_metal"]


def test_cpu_tensors_fail_loudly_no_fallback():
    x = torch.zeros(1, 4, 4, 3)
    with pytest.raises(RuntimeError, match="ROCm device"):
        matching.match_simple(x, x)
    with pytest.raises(RuntimeError, match="ROCm device"):
        kernels.iter_proj(torch.zeros(1, 4, 4, 9), torch.zeros(1, 16, 3), torch.zeros(1, 16, 2))
    with pytest.raises(RuntimeError, match="ROCm device"):
        tracker.opt_pose_ray_dist_sim3(torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(8), torch.zeros(8),
                                       torch.ones(4), torch.ones(4))


def test_product_package_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mast3r-slam_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_config_defaults_and_merge():
    c = config.get_config()
    m, t = c["matching"], c["tracking"]
    assert (m["max_iter"], m["lambda_init"], m["convergence_thresh"], m["dist_thresh"]) == (10, 1e-8, 1e-6, 0.1)
    assert (m["refine_radius"], m["refine_dilation"], m["use_simple"]) == (3, 2, True)
    assert (t["max_iters"], t["huber"], t["sigma_ray"], t["sigma_dist"], t["Q_conf"]) == (10, 1.345, 0.003, 10.0, 1.5)
    config.set_config({"matching": {"use_simple": False}})
    try:
        assert config.get_config()["matching"]["use_simple"] is False
        assert config.get_config()["matching"]["max_iter"] == 10
    finally:
        config.reset_config()
    assert config.get_config()["matching"]["use_simple"] is True


def test_default_config_equals_the_reference_constants(golden_dir):
    """tests/golden/reference_default_config.json = the reference's DEFAULT_CONFIG (config.py:53-) dumped by
    oracle/make_golden.py.  Every constant the hot path keeps must have the reference's value; what this repo ADDS is
    listed here, with where its default comes from."""
    import json
    ref = json.load(open(os.path.join(golden_dir, "reference_default_config.json")))
    ours = config.DEFAULT_CONFIG
    added = {
        ("matching", "refine_radius"), ("matching", "refine_dilation"), ("matching", "use_refine"),   # matching.py:405-407's .get() defaults
        ("matching", "refine_chained"),                                                              # numpy-twin vs Metal semantics
        ("matching", "use_fast_nn"), ("matching", "fast_nn_subsample"), ("matching", "fast_nn_rounds"),   # K8: not in the reference
        ("local_opt", "min_match_frac"),                                                             # slam.py:309-311's .get() default
    }
    seen = 0
    for sec, val in ours.items():
        if isinstance(val, dict):
            for k, v in val.items():
                if (sec, k) in added:
                    assert k not in ref.get(sec, {}), (sec, k)
                    continue
                assert sec in ref and k in ref[sec], (sec, k)
                assert ref[sec][k] == v, (sec, k, ref[sec][k], v)
                seen += 1
        else:
            assert ref[sec] == val, sec
            seen += 1
    assert seen >= 40
    assert ours["local_opt"]["min_match_frac"] == 0.1 and ours["reloc"]["min_match_frac"] == ref["reloc"]["min_match_frac"] == 0.3


def test_local_map_follows_reference_pinning():
    uniq, local, nfree = kernels._local_map(np.array([3, 1, 1]), np.array([1, 4, 3]), 6, pin=1)
    assert uniq.tolist() == [1, 3, 4] and nfree == 2
    assert local.tolist() == [-1, -1, -1, 0, 1, -1]


def test_factor_graph_edge_bookkeeping_cpu():
    """global_opt.py:140-166 host logic (pure torch, runs on CPU tensors): two-way edges in the reference's
    order, and edges re-indexed to rows of the unique-keyframe stack when keyframe ids are not 0..K-1."""
    import torch
    from types import SimpleNamespace
    from mast3r_slam.global_opt import FactorGraph
    fg = FactorGraph(SimpleNamespace(device="cpu"), frames=[])
    n = 6
    fg.ii, fg.jj = torch.tensor([2, 5], dtype=torch.int32), torch.tensor([5, 9], dtype=torch.int32)
    fg.owner = torch.zeros(2, dtype=torch.int32)                  # no group: every edge is stored here
    fg.idx_ii2jj = torch.arange(2 * n).reshape(2, n)
    fg.idx_jj2ii = 100 + torch.arange(2 * n).reshape(2, n)
    fg.valid_match_j = torch.ones(2, n, 1, dtype=torch.bool)
    fg.valid_match_i = torch.zeros(2, n, 1, dtype=torch.bool)
    fg.Q_ii2jj, fg.Q_jj2ii = torch.full((2, n, 1), 2.0), torch.full((2, n, 1), 3.0)
    uniq = fg.get_unique_kf_idx()
    assert uniq.tolist() == [2, 5, 9]
    ii, jj, idx, valid, Q = fg._prep_two_way_edges()
    assert ii.tolist() == [2, 5, 5, 9] and jj.tolist() == [5, 9, 2, 5]
    assert idx[2, 0] == 100 and valid[:2].all() and not valid[2:].any() and Q[3, 0, 0] == 3.0
    li, lj, lidx, lvalid, lQ, _ = fg._local_edges(uniq)
    assert li.tolist() == [0, 1, 1, 2] and lj.tolist() == [1, 2, 0, 1]
    assert lidx.dtype == torch.int32 and lvalid.shape == (4, n) and lQ.shape == (4, n)


def test_resize_img_contract():
    """mast3r_utils.py:132-207: long edge 512, crop to multiples of 16, square -> 4:3 unless square_ok;
    224: short edge 224 + centre square crop; normalisation (x/255 - 0.5)/0.5; transformation tuple."""
    import numpy as np
    from mast3r_slam.mast3r_utils import resize_img
    r = np.random.default_rng(0)
    img = r.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    out, (sw, sh, cw, ch) = resize_img(img, 512, return_transformation=True)
    assert tuple(out["img"].shape) == (1, 384, 512, 3) and out["true_shape"].tolist() == [[384, 512]]
    assert abs(sw - 640 / 512) < 1e-9 and abs(sh - 480 / 384) < 1e-9 and (cw, ch) == (0, 0)
    raw = out["unnormalized_img"]
    assert raw.dtype == np.uint8 and np.allclose(out["img"][0].numpy(), (raw / 255.0 - 0.5) / 0.5, atol=1e-6)
    sq = r.random((600, 600, 3)).astype(np.float32)                          # float [0,1] input, upsampling not needed
    assert tuple(resize_img(sq, 512)["img"].shape) == (1, 384, 512, 3)       # square -> 4:3
    assert tuple(resize_img(sq, 512, square_ok=True)["img"].shape) == (1, 512, 512, 3)
    odd = r.integers(0, 256, size=(333, 500, 3), dtype=np.uint8)             # 512 x 341 after resize -> 336 rows
    o = resize_img(odd, 512)
    h, w = o["true_shape"][0].tolist()
    assert (h % 16, w % 16) == (0, 0) and (h, w) == (336, 512)
    assert tuple(resize_img(img, 224)["img"].shape) == (1, 224, 224, 3)
    small = r.integers(0, 256, size=(120, 160, 3), dtype=np.uint8)           # enlarging path (BICUBIC)
    assert tuple(resize_img(small, 512)["img"].shape) == (1, 384, 512, 3)


def test_checkpoint_loader_layouts_cpu(tmp_path):
    """model.load_state_dict (behind Mast3rFull.from_pretrained(weights_path=...), reference call
    mast3r_utils.py:67-76): bare state dict, the public checkpoint wrapper {"model": ..., "args": Namespace}
    (which torch.load's weights_only default refuses), safetensors; unrelated files are rejected."""
    import argparse

    import torch

    from mast3r_slam import model as M
    g = torch.Generator().manual_seed(3)                        # a few tensors with public key names are enough here
    w = {"patch_embed.proj.weight": torch.randn(8, 3, 16, 16, generator=g), "patch_embed.proj.bias": torch.randn(8, generator=g),
         "enc_blocks.0.attn.qkv.weight": torch.randn(24, 8, generator=g), "dec_norm.weight": torch.randn(8, generator=g)}
    torch.save(w, tmp_path / "bare.pth")
    torch.save({"model": dict(w, mask_token=torch.zeros(1, 1, 768).half()), "args": argparse.Namespace(x=1), "epoch": 7},
               tmp_path / "ckpt.pth")
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in w.items()}, str(tmp_path / "w.safetensors"))
    for name in ("bare.pth", "ckpt.pth", "w.safetensors"):
        sd = M.load_state_dict(str(tmp_path / name))
        assert all(torch.equal(sd[k], v) and sd[k].dtype == torch.float32 for k, v in w.items()), name
    torch.save({"foo": torch.zeros(2)}, tmp_path / "bad.pth")
    with pytest.raises(KeyError, match="not a MASt3R state dict"):
        M.load_state_dict(str(tmp_path / "bad.pth"))


def test_precision_argument_is_validated_before_the_device_is_needed():
    """load_mast3r's precision (mast3r_utils.py:51: "fp16" | "fp32" | "bf16"): an unknown value is a ValueError
    even on a box without a GPU; a known one only then asks for the device (RuntimeError here, no CPU path)."""
    import torch

    from mast3r_slam import mast3r_utils
    with pytest.raises(ValueError, match="precision"):
        mast3r_utils.load_mast3r("mast3r_full", precision="int8")
    # no checkpoint path: the reference would download one; here that is an error, never silent random weights
    with pytest.raises(FileNotFoundError, match="random_init=True"):
        mast3r_utils.load_mast3r("mast3r_full", precision="bf16")
    if not torch.cuda.is_available():
        for prec in ("bf16", "fp16", "fp32"):
            with pytest.raises(RuntimeError, match="ROCm device"):
                mast3r_utils.load_mast3r("mast3r_full", precision=prec, random_init=True)
