"""GPU parity of the two-view network at the depth and sizes it ships and is benchmarked at
(BASELINE configs[1]: ViT-L encoder 24 blocks + 2 x 12 decoder blocks, 512x512; configs[0]: 512x384), at the
non-multiple-of-128 token counts resize_img emits, the checkpoint loader, the `precision` argument and the
symmetric (ii, ji, jj, ij) operator - all against the torch-CPU fp32 oracle (oracle/model.py) and the numpy
matcher oracle.  Contract replaced: model.reconstruct -> pts3d / conf / desc / desc_conf
(/root/reference/src/mlx_mast3r_slam/mast3r_utils.py:281-294, :355), tolerance of BASELINE.json: 1e-3 rel-L2."""
import argparse
import os

import numpy as np
import pytest
import torch

from mast3r_slam import config, mast3r_utils, matching as matching_mod, model as M, synthetic
from mast3r_slam.frame import create_frame
from oracle import matching as om
from oracle import model as OM

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())


@pytest.fixture(scope="module")
def full_weights():
    return M.init_random_weights(M.FULL_CFG, seed=0)


def _pair(h, w, s0):
    return synthetic.textured_image(h, w, s0)[None], synthetic.textured_image(h, w, s0 + 1)[None]


_ORACLE = {}


def _oracle_pair(weights, family, h, w, s0=0):
    """fp32 oracle outputs of the pair (s0, s0 + 1) at h x w, computed once per module run (~10 s of CPU at full depth)."""
    key = (family, h, w, s0)
    if key not in _ORACLE:
        im1, im2 = _pair(h, w, s0)
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        with torch.no_grad():
            _ORACLE[key] = OM.reconstruct(weights, torch.from_numpy(im1), torch.from_numpy(im2), M.FULL_CFG)
    return _ORACLE[key]


# Tolerances per precision mode, rel-L2 against the fp32 oracle at FULL depth.  A CPU emulation of operand
# rounding predicts pts3d 1.0e-3 / 4e-4 / 1.3e-4 for all-bf16 / bf16 trunk + fp16 heads /
# fp16; the shipped default ("bf16") must meet BASELINE.json's 1e-3.
FULL_TOL = {
    "bf16": dict(pts3d=1e-3, conf=1e-3, desc=6e-3, desc_conf=6e-3),
    "fp16": dict(pts3d=4e-4, conf=1e-4, desc=1.5e-3, desc_conf=1.5e-3),
}


@pytest.mark.parametrize("shape", [(512, 512), (384, 512)])
def test_full_depth_network_vs_cpu_oracle(dev, full_weights, shape):
    """FULL_CFG (24 + 12 + 12 blocks), one pair: every output of both views within the stated tolerance of the fp32
    oracle, in the shipped default precision ("bf16": bf16 trunk, fp16 heads) and in "fp16" (the reference's
    default precision, mast3r_utils.py:51).  The all-bf16 variant is measured beside them and must stay under
    3e-3: it is what the fp16 heads buy."""
    h, w = shape
    im1, im2 = _pair(h, w, 0)
    r1, r2 = _oracle_pair(full_weights, "plain", h, w)
    report = {}
    for prec, kw in (("bf16", {}), ("fp16", {}), ("bf16", dict(head_precision="bf16"))):
        net = M.Mast3rFull(weights=full_weights, device=dev, precision=prec, **kw)
        o1, o2 = net.reconstruct_batch(im1, im2)
        name = prec + ("+bf16heads" if kw else "")
        for v, (o, r) in enumerate(((o1, r1), (o2, r2))):
            assert o["pts3d"].shape == (1, h, w, 3) and o["desc"].shape == (1, h, w, 24)
            errs = {k: _rel(o[k], r[k]) for k in ("pts3d", "conf", "desc", "desc_conf")}
            report[(name, v)] = errs
            assert all(torch.isfinite(o[k]).all() for k in errs)
            if kw:
                assert errs["pts3d"] < 3e-3 and errs["desc"] < 8e-3, (name, v, errs)
            else:
                for k, tol in FULL_TOL[prec].items():
                    assert errs[k] < tol, (name, v, k, errs)
        del net
        torch.cuda.empty_cache()
    print("\nfull-depth rel-L2 vs fp32 oracle", shape, {k: {n: f"{e:.2e}" for n, e in v.items()} for k, v in report.items()})
    # the fp16 heads must actually pay: pointmap error below the all-bf16 variant in both views
    for v in range(2):
        assert report[("bf16", v)]["pts3d"] < report[("bf16+bf16heads", v)]["pts3d"]


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_benchmarked_configuration_eight_pairs_through_the_graph(dev, full_weights, prec):
    """What bench.py times (BASELINE configs[3]'s per-GPU shard): FULL_CFG, 8 pairs at 512x512, bench.py's default precision
    ("fp16": the LayerNorm-folded trunk) and the bf16 trunk BASELINE configs[1] names,
    replayed from the hipGraph of `net.graphed(8, 512, 512)` - i.e. the k_gemm256 256x256 / 256x192 tiles, the 2-group
    launches and the XCD-ordered grids, none of which a one-pair eager pass reaches (it runs on the 64 / 128 tiles).
    Contract: mast3r_utils.py:281-294.
      * every output of every pair equals the same pair run ALONE, eagerly, bit for bit ("same bits alone or in a
        batch": all tile shapes accumulate K in the same order, split-K slice counts depend on the per-image geometry
        only, attention works per (image, head));
      * pair 0 is within FULL_TOL of the fp32 oracle (hence every pair of the batch is as good as the one-pair test says)."""
    h = w = 512
    P = 8
    im1 = np.stack([synthetic.textured_image(h, w, 2 * p) for p in range(P)])          # bench.py's images of rank 0
    im2 = np.stack([synthetic.textured_image(h, w, 2 * p + 1) for p in range(P)])
    net = M.Mast3rFull(weights=full_weights, device=dev, precision=prec)
    g = net.graphed(P, h, w)
    o1, o2 = g(im1, im2)
    o1, o2 = ({k: v.clone() for k, v in o.items()} for o in (o1, o2))
    q1, q2 = g(im1, im2)                                                                # a second replay: same bits
    keys = ("pts3d", "conf", "desc", "desc_conf")
    for k in keys:
        assert torch.equal(q1[k], o1[k]) and torch.equal(q2[k], o2[k]), k
    for p in range(P):
        e1, e2 = net.reconstruct_batch(im1[p:p + 1], im2[p:p + 1])
        for k in keys:
            assert torch.equal(o1[k][p], e1[k][0]), (p, k, "view 1", _rel(o1[k][p], e1[k][0]))
            assert torch.equal(o2[k][p], e2[k][0]), (p, k, "view 2", _rel(o2[k][p], e2[k][0]))
    r1, r2 = _oracle_pair(full_weights, "plain", h, w)
    for o, r in ((o1, r1), (o2, r2)):
        for k, tol in FULL_TOL[prec].items():
            assert _rel(o[k][0], r[k][0]) < tol, (k, _rel(o[k][0], r[k][0]))
    assert not torch.equal(o1["pts3d"][0], o1["pts3d"][1])                             # the pairs really differ


@pytest.mark.parametrize("prec,P,shape", [("fp16", 3, (384, 512)), ("bf16", 2, (336, 512))])
def test_batch_versus_alone_at_other_precisions_sizes_and_pair_counts(dev, full_weights, prec, P, shape):
    """The same "alone == in a batch" statement where the DISPATCH differs from the benchmarked case: the loader's default
    precision (fp16 operands with bf16 P.V attention, k_attn<.., DT_F16, 2, DT_BF16>), BASELINE configs[0]'s 512x384
    (768 tokens: 6 key tiles, 192-row tails) and a 672-token size resize_img emits, odd pair counts (3 pairs = 6 images:
    the direct convolutions switch on for some maps and not for others, the dense launches fall between the 128 and
    the 256 tiles).  Full depth, graph replay of P pairs against every pair alone, bit for bit."""
    h, w = shape
    im1 = np.stack([synthetic.textured_image(h, w, 20 + 2 * p) for p in range(P)])
    im2 = np.stack([synthetic.textured_image(h, w, 21 + 2 * p) for p in range(P)])
    net = M.Mast3rFull(weights=full_weights, device=dev, precision=prec)
    o1, o2 = net.graphed(P, h, w)(im1, im2)
    o1, o2 = ({k: v.clone() for k, v in o.items()} for o in (o1, o2))
    for p in range(P):
        e1, e2 = net.reconstruct_batch(im1[p:p + 1], im2[p:p + 1])
        for k in ("pts3d", "conf", "desc", "desc_conf"):
            assert torch.isfinite(o1[k][p]).all() and torch.isfinite(o2[k][p]).all(), (p, k)
            assert torch.equal(o1[k][p], e1[k][0]), (prec, p, k, "view 1", _rel(o1[k][p], e1[k][0]))
            assert torch.equal(o2[k][p], e2[k][0]), (prec, p, k, "view 2", _rel(o2[k][p], e2[k][0]))
    assert not torch.equal(o1["pts3d"][0], o1["pts3d"][1])


# Trained-like statistics (model.init_random_weights(family="trained_like")): rel-L2 bounds at FULL depth, 512x512.
# CPU emulation of operand rounding (tools/emul_precision.py --family trained_like --res 512 512): pts3d view 1 / view 2
#   bf16 trunk + fp16 heads 1.28e-3 / 2.34e-3 (encoder GEMM operands 9e-4 / 1.7e-3, encoder q|k|v|P 9e-4 / 1.6e-3: an
#   8-bit mantissa on logits of 30-80 moves softmax weights by percents), fp16 1.8e-4 / 3.4e-4.
TRAINED_TOL = {
    "fp16": dict(pts3d=1e-3, conf=1e-4, desc=6e-3, desc_conf=8e-3),      # BASELINE.json's 1e-3 is met by the fp16 trunk
    "bf16": dict(pts3d=5e-3, conf=2e-4, desc=3e-2, desc_conf=4e-2),      # stated bound of the bf16 trunk: it does NOT meet 1e-3 here
}


def test_trained_like_weight_statistics_at_full_depth(dev):
    """The plain random family is a near-identity network (uniform softmax, no outlier channels) and flatters 16-bit
    operands.  This family has what trained ViT-L checkpoints have - LayerNorm gains over a decade, softmax rows that
    peak (row maxima of 20-80), massive-activation channels 50x the stream's median, GELU inputs beyond 6 - and both
    precision modes are held to a written bound at the shipped depth and size:
      * precision="fp16" (load_mast3r's default, as the reference's): pts3d < 1e-3 in both views;
      * precision="bf16" (BASELINE configs[1]): < 5e-3 - measured 1.3e-3 / 2.3e-3, i.e. it misses 1e-3 on such weights,
        which is why it is not the loader's default.
    Also exercises the deferred-maximum attention loop (MODE 2) on scores that outgrow the first tile's maximum."""
    h = w = 512
    wt = M.init_random_weights(M.FULL_CFG, seed=0, family="trained_like")
    im1, im2 = _pair(h, w, 0)
    r1, r2 = _oracle_pair(wt, "trained_like", h, w)
    report = {}
    for prec in ("fp16", "bf16"):
        net = M.Mast3rFull(weights=wt, device=dev, precision=prec)
        o1, o2 = net.reconstruct_batch(im1, im2)
        for v, (o, r) in enumerate(((o1, r1), (o2, r2))):
            errs = {k: _rel(o[k], r[k]) for k in ("pts3d", "conf", "desc", "desc_conf")}
            report[(prec, v)] = errs
            assert all(torch.isfinite(o[k]).all() for k in errs), (prec, v)
            for k, tol in TRAINED_TOL[prec].items():
                assert errs[k] < tol, (prec, v, k, errs)
        del net
        torch.cuda.empty_cache()
    print("\ntrained-like family, full depth, rel-L2 vs fp32 oracle", {k: {n: f"{e:.2e}" for n, e in v.items()} for k, v in report.items()})
    for v in range(2):
        assert report[("fp16", v)]["pts3d"] < 0.5 * report[("bf16", v)]["pts3d"]       # the 3 extra mantissa bits pay


@pytest.fixture(scope="module", params=["bf16", "fp16"])
def tiny(dev, request):
    """Both trunk precisions: "fp16" (load_mast3r's default) runs the LayerNorm fold (m3_gemm_ex) - ragged token counts,
    cached-token decodes and the operator API all go through it; "bf16" keeps the LayerNorm kernels."""
    cfg = M.TINY_CFG
    w = M.init_random_weights(cfg, seed=1)
    return cfg, w, M.Mast3rFull(weights=w, cfg=cfg, device=dev, precision=request.param)


@pytest.mark.parametrize("shape", [(336, 512), (288, 512), (224, 224), (512, 336)])
def test_token_counts_that_are_not_multiples_of_128(tiny, dev, shape):
    """resize_img (mast3r_utils.py:132-207) emits any multiple of 16: 512x336 -> T = 672, 16:9 video -> 512x288 ->
    T = 576, 224x224 -> 196.  Two pairs at once (batch > 1 crosses the image boundary inside a GEMM row tile)."""
    cfg, w, net = tiny
    h, wd = shape
    im1 = np.stack([synthetic.textured_image(h, wd, s) for s in (0, 2)])
    im2 = np.stack([synthetic.textured_image(h, wd, s) for s in (1, 3)])
    o1, o2 = net.reconstruct_batch(im1, im2)
    with torch.no_grad():
        r1, r2 = OM.reconstruct(w, torch.from_numpy(im1), torch.from_numpy(im2), cfg)
    for o, r in ((o1, r1), (o2, r2)):
        assert o["pts3d"].shape == (2, h, wd, 3)
        assert _rel(o["pts3d"], r["pts3d"]) < 1e-3 and _rel(o["conf"], r["conf"]) < 1e-4
        assert _rel(o["desc"], r["desc"]) < 6e-3 and _rel(o["desc_conf"], r["desc_conf"]) < 6e-3
    g = net.graphed(1, h, wd)                                       # the captured path accepts the shape too
    e1, _ = net.reconstruct_batch(im1[:1], im2[:1])
    q1, _ = g(im1[:1], im2[:1])
    assert torch.equal(q1["pts3d"], e1["pts3d"])


def test_odd_token_grid_one_image_and_one_pair(tiny, dev):
    """336 x 336 -> a 21 x 21 grid, 441 tokens: model.encode of ONE image is a 441-row stream and one pair decodes two 441-row
    groups - odd row counts, which the LayerNorm fold (row pairs) hands to the LayerNorm kernels.  Same contract and the same
    1e-3 against the oracle as every other shape."""
    cfg, w, net = tiny
    h = wd = 336
    im1, im2 = _pair(h, wd, 5)
    tok = net.encode(im1[0])
    assert tok.shape == (441, cfg["enc_dim"]) and torch.isfinite(tok.float()).all()
    both = net.encode(np.concatenate([im1, im2]))                      # 882 rows: the folded path on the fp16 trunk
    assert _rel(tok, both[0]) < 3e-3
    o1, o2 = net.reconstruct_batch(im1, im2)
    with torch.no_grad():
        r1, r2 = OM.reconstruct(w, torch.from_numpy(im1), torch.from_numpy(im2), cfg)
    for o, r in ((o1, r1), (o2, r2)):
        assert o["pts3d"].shape == (1, h, wd, 3)
        assert _rel(o["pts3d"], r["pts3d"]) < 1e-3 and _rel(o["conf"], r["conf"]) < 1e-4


def test_checkpoint_round_trip_and_precision_argument(tiny, dev, tmp_path):
    """load_mast3r(model_type, variant, resolution, precision) + from_pretrained(weights_path)
    (mast3r_utils.py:47-80, :67-76): a saved state dict - bare, wrapped as the public checkpoint
    {"model": sd, "args": Namespace}, and .safetensors - reproduces the in-memory model bit for bit;
    precision accepts "bf16" | "fp16" | "fp32" (reference signature) and rejects anything else."""
    cfg, w, net = tiny
    h, wd = 128, 256
    im1, im2 = _pair(h, wd, 0)
    ref1, ref2 = net.reconstruct_batch(im1, im2)
    bare, wrapped, st = tmp_path / "bare.pth", tmp_path / "ckpt.pth", tmp_path / "w.safetensors"
    torch.save(w, bare)
    torch.save({"model": dict(w, mask_token=torch.zeros(1, 1, 768)), "args": argparse.Namespace(model="AsymmetricMASt3R(...)")}, wrapped)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in w.items()}, str(st))
    for path in (bare, wrapped, st):
        m2 = mast3r_utils.load_mast3r("mast3r_full", "base", 512, net.precision, weights_path=str(path), cfg=cfg, device=dev)
        o1, o2 = m2.reconstruct_batch(im1, im2)
        for k in ref1:
            assert torch.equal(o1[k], ref1[k]) and torch.equal(o2[k], ref2[k]), (path.name, k)
    torch.save({"something": torch.zeros(3)}, tmp_path / "bad.pth")
    with pytest.raises(KeyError, match="not a MASt3R state dict"):
        M.Mast3rFull.from_pretrained(weights_path=str(tmp_path / "bad.pth"), cfg=cfg, device=dev)
    with torch.no_grad():
        r1, _ = OM.reconstruct(w, torch.from_numpy(im1), torch.from_numpy(im2), cfg)
    errs = {}
    for prec in ("bf16", "fp16", "fp32"):
        m3 = mast3r_utils.load_mast3r("mast3r_full", "base", 512, prec, weights_path=str(bare), cfg=cfg, device=dev)
        assert m3.encode(im1[0]).dtype == (torch.bfloat16 if prec == "bf16" else torch.float16)
        errs[prec] = _rel(m3.reconstruct_batch(im1, im2)[0]["pts3d"], r1["pts3d"])
        assert errs[prec] < 1e-3
    assert errs["fp16"] < errs["bf16"] and errs["fp32"] == errs["fp16"]     # "fp32" is served by fp16 operands (documented)
    with pytest.raises(ValueError, match="precision"):
        mast3r_utils.load_mast3r("mast3r_full", precision="fp8", cfg=cfg, device=dev)


def test_symmetric_operators_against_the_oracles(tiny, dev):
    """mast3r_symmetric_inference / mast3r_decode_symmetric_batch / mast3r_match_symmetric (mast3r_utils.py:382-443,
    :503-632): the (ii, ji, jj, ij) order against the ORACLE network run on (i, j) and on the swapped pair (j, i),
    and both matching directions bit-exact against oracle.matching.match_iterative_proj on the swapped inputs."""
    cfg, w, net = tiny
    h, wd = 128, 256
    n = h * wd
    imi, imj = synthetic.textured_image(h, wd, 10), synthetic.textured_image(h, wd, 11)
    fi = create_frame(0, torch.from_numpy(imi).to(dev))
    fj = create_frame(1, torch.from_numpy(imj).to(dev))
    X4, C4, D4, Q4 = mast3r_utils.mast3r_symmetric_inference(net, fi, fj)
    ti, tj = torch.from_numpy(imi)[None], torch.from_numpy(imj)[None]
    with torch.no_grad():
        (rii, rji), (rjj, rij) = OM.reconstruct(w, ti, tj, cfg), OM.reconstruct(w, tj, ti, cfg)
    for k, r in enumerate((rii, rji, rjj, rij)):                     # order (ii, ji, jj, ij), mast3r_utils.py:424
        assert _rel(X4[k], r["pts3d"][0]) < 1e-3 and _rel(D4[k], r["desc"][0]) < 6e-3, k
        assert _rel(C4[k], r["conf"][0]) < 1e-4 and _rel(Q4[k], r["desc_conf"][0]) < 6e-3, k
    assert _rel(X4[0], rjj["pts3d"][0]) > 1e-2                       # ... and the views really differ
    feats_i, feats_j = fi.feat[None], fj.feat[None]
    shp = [torch.tensor([[h, wd]])]
    config.set_config({"matching": {"use_simple": False}})
    try:
        Xs, Cs, Ds, Qs = mast3r_utils.mast3r_decode_symmetric_batch(net, feats_i, None, feats_j, None, shp, shp)
        idx_i2j, idx_j2i, valid_j, valid_i, Qii, Qjj, Qji, Qij = mast3r_utils.mast3r_match_symmetric(
            net, feats_i, None, feats_j, None, shp, shp)
    finally:
        config.reset_config()
    for k in range(4):
        assert torch.equal(Xs[k, 0], X4[k]) and torch.equal(Ds[k, 0], D4[k])
    c = lambda t: np.ascontiguousarray(t.cpu().numpy())
    # i -> j: match(X11 = ii, X21 = ji); j -> i: the same call on the swapped pair, (jj, ij)
    io, vo = om.match_iterative_proj(c(Xs[0]), c(Xs[1]), c(Ds[0]), c(Ds[1]), dilation_max=2)
    assert np.array_equal(c(idx_i2j), io) and np.array_equal(c(valid_j), vo)
    io, vo = om.match_iterative_proj(c(Xs[2]), c(Xs[3]), c(Ds[2]), c(Ds[3]), dilation_max=2)
    assert np.array_equal(c(idx_j2i), io) and np.array_equal(c(valid_i), vo)
    assert not np.array_equal(c(idx_i2j), c(idx_j2i))
    for got, k in ((Qii, 0), (Qjj, 2), (Qji, 1), (Qij, 3)):           # return order :533 = (Qii, Qjj, Qji, Qij)
        assert torch.equal(got, Qs[k].reshape(1, n, 1))


def test_grouped_heads_equal_separate_heads(tiny, dev):
    """Mast3rFull.heads (every head operator as one 2-group launch, no side stream) returns the same bits as two
    Mast3rFull.head calls: all tile shapes accumulate K in the same order, split-K slices depend on the per-image
    geometry only, and the grouped kernels only add a blockIdx.y -> weights / offsets selection."""
    cfg, w, net = tiny
    h, wd = 128, 256
    im1, im2 = _pair(h, wd, 4)
    tok, grid = net.encode_tokens(net._as_images(np.concatenate([im1, im2], 0)))
    m = grid[0] * grid[1]
    taps = net.decode_tokens(tok[:m], tok[m:], 1, grid)
    g1, g2 = net.heads(taps[0], taps[1], 1, grid)
    s1, s2 = net.head("downstream_head1", taps[0], 1, grid), net.head("downstream_head2", taps[1], 1, grid)
    for g, s in ((g1, s1), (g2, s2)):
        for k in s:
            assert torch.equal(g[k], s[k]), k
    # and a caller that captures on a FORKED stream (the nested-fork capture that used to need concurrent_heads=False)
    side = torch.cuda.Stream()
    a = torch.from_numpy(im1).to(dev); b = torch.from_numpy(im2).to(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            o_side = net.reconstruct_batch(b, a)
        o_main = net.reconstruct_batch(a, b)
        cur.wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    e_main, e_side = net.reconstruct_batch(a, b), net.reconstruct_batch(b, a)
    for v in range(2):
        for k in e_main[v]:
            assert torch.equal(o_main[v][k], e_main[v][k]) and torch.equal(o_side[v][k], e_side[v][k])


def test_fp16_features_through_the_operator_api(tiny, dev):
    """BASELINE configs[4] "fp16 features": a model built with features="fp16" emits half descriptors (the fp32 value
    rounded once); the symmetric match operator on them equals the numpy oracle matcher run on the half-rounded
    descriptors bit for bit, and everything else the network returns is unchanged."""
    cfg, w, net = tiny
    net16 = M.Mast3rFull(weights=w, cfg=cfg, device=dev, features="fp16", precision=net.precision)
    with pytest.raises(ValueError, match="features"):
        M.Mast3rFull(weights=w, cfg=cfg, device=dev, features="int8")
    h, wd = 128, 256
    imi, imj = synthetic.textured_image(h, wd, 10), synthetic.textured_image(h, wd, 11)
    fi = create_frame(0, torch.from_numpy(imi).to(dev)); fj = create_frame(1, torch.from_numpy(imj).to(dev))
    gi = create_frame(0, torch.from_numpy(imi).to(dev)); gj = create_frame(1, torch.from_numpy(imj).to(dev))
    X4, C4, D4, Q4 = mast3r_utils.mast3r_symmetric_inference(net, fi, fj)
    Xh, Ch, Dh, Qh = mast3r_utils.mast3r_symmetric_inference(net16, gi, gj)
    assert Dh.dtype == torch.float16 and torch.equal(Dh, D4.half())
    assert torch.equal(Xh, X4) and torch.equal(Ch, C4) and torch.equal(Qh, Q4)
    shp = [torch.tensor([[h, wd]])]
    config.set_config({"matching": {"use_simple": False}})
    try:
        idx_i2j, idx_j2i, valid_j, valid_i, *_ = mast3r_utils.mast3r_match_symmetric(net16, gi.feat[None], None, gj.feat[None],
                                                                                      None, shp, shp)
    finally:
        config.reset_config()
    c = lambda t: np.ascontiguousarray(t.float().cpu().numpy())
    io, vo = om.match_iterative_proj(c(Xh[0])[None], c(Xh[1])[None], c(Dh[0])[None], c(Dh[1])[None], dilation_max=2)
    assert np.array_equal(idx_i2j.cpu().numpy(), io) and np.array_equal(valid_j.cpu().numpy(), vo)
    io, vo = om.match_iterative_proj(c(Xh[2])[None], c(Xh[3])[None], c(Dh[2])[None], c(Dh[3])[None], dilation_max=2)
    assert np.array_equal(idx_j2i.cpu().numpy(), io) and np.array_equal(valid_i.cpu().numpy(), vo)


def test_match_operators_with_the_fast_reciprocal_nn_matcher(tiny, dev):
    """matching.use_fast_nn behind the operator API (mast3r_match_asymmetric / _symmetric, mast3r_utils.py:451-533's
    contract): same shapes and dtypes as with the dense matchers, at most one valid match per seed, every valid index is
    the reciprocal nearest neighbour fast_reciprocal_nn_maps reports for the network's own descriptors (random weights:
    maps without spatial coherence, i.e. the block-bound search hands over to its brute-force fallback on the device)."""
    cfg, w, net = tiny
    h, wd = 128, 256
    n = h * wd
    imi, imj = synthetic.textured_image(h, wd, 20), synthetic.textured_image(h, wd, 21)
    fi = create_frame(0, torch.from_numpy(imi).to(dev)); fj = create_frame(1, torch.from_numpy(imj).to(dev))
    X, C, D, Q = mast3r_utils.mast3r_asymmetric_inference(net, fi, fj)
    config.set_config({"matching": {"use_fast_nn": True, "fast_nn_subsample": 8, "fast_nn_rounds": 3, "dist_thresh": 1e9}})
    try:
        idx, valid, Xii, Cii, Qii, Xji, Cji, Qji = mast3r_utils.mast3r_match_asymmetric(net, fi, fj)
        shp = [torch.tensor([[h, wd]])]
        s_ij, s_ji, v_j, v_i, *_ = mast3r_utils.mast3r_match_symmetric(net, fi.feat[None], None, fj.feat[None], None, shp, shp)
    finally:
        config.reset_config()
    seeds = (h // 8) * (wd // 8)
    assert idx.shape == (1, n) and idx.dtype == torch.int64 and valid.shape == (1, n, 1) and valid.dtype == torch.bool
    assert Xii.shape == (1, n, 3) and Qji.shape == (1, n, 1)
    assert 0 < int(valid.sum()) <= seeds
    m = matching_mod.fast_reciprocal_nn_maps(D[0][None], D[1][None], subsample=8, max_iter=3)
    assert torch.equal(valid[0, :, 0], m["valid"][0, :, 0])                      # dist_thresh switched off above
    assert torch.equal(idx[0][valid[0, :, 0]], m["idx"][0][valid[0, :, 0]])
    for s_, v_ in ((s_ij, v_j), (s_ji, v_i)):
        assert s_.shape == (1, n) and v_.shape == (1, n, 1) and 0 < int(v_.sum()) <= seeds
        assert int(s_.min()) >= 0 and int(s_.max()) < n
