"""GPU parity of the nearest-neighbour search (m3_nn_search through the C ABI) and of fast reciprocal NN matching
against the float64 oracle (oracle/matching.py).  The device scores are an fp32 FMA chain, so an index must equal
the oracle's wherever the float64 margin between the best and the runner-up exceeds the fp32 bound (D * 2^-23 for
unit vectors), and must be score-optimal within that bound everywhere."""
import numpy as np
import pytest
import torch

from mast3r_slam import matching, synthetic
from oracle import matching as om

pytestmark = pytest.mark.gpu
EPS = 24 * 2.0 ** -23


def _unit(rng, n, d):
    v = rng.normal(size=(n, d)).astype(np.float32)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


@pytest.mark.parametrize("method", ["mfma", "fma"])
@pytest.mark.parametrize("s,n,d", [(1, 1, 24), (300, 5000, 24), (4096, 16384, 24), (257, 777, 16), (64, 1000, 32), (600, 70001, 24)])
def test_nn_search_vs_float64_oracle(dev, s, n, d, method):
    rng = np.random.default_rng(s + n)
    Q, DB = _unit(rng, s, d), _unit(rng, n, d)
    DB[n // 2] = DB[0]                                                       # an exact duplicate: ties go to the lowest index
    idx, score = matching.nn_search(torch.from_numpy(Q)[None].to(dev), torch.from_numpy(DB)[None].to(dev), return_score=True,
                                    method=method)
    idx, score = idx[0].cpu().numpy(), score[0].cpu().numpy()
    ref, best, second = om.nn_search(Q, DB)
    clear = best - second > 4 * EPS
    assert np.array_equal(idx[clear], ref[clear]) and clear.mean() > 0.9 or n < 10
    chosen = np.einsum("sd,sd->s", Q.astype(np.float64), DB[idx].astype(np.float64))
    assert np.all(chosen >= best - 4 * EPS) and np.abs(score - chosen).max() < 4 * EPS
    dup = ref == 0
    assert np.all(idx[dup & clear] == 0)                                     # never the duplicate at n // 2


def test_nn_search_fp16_descriptors_are_exact_products(dev):
    """Descriptors STORED as fp16 (BASELINE configs[4] 'fp16 features'): the MFMA multiplies them exactly and sums in
    fp32, so against the float64 oracle on the SAME fp16 values only the fp32 summation bound applies."""
    rng = np.random.default_rng(11)
    Q, DB = _unit(rng, 1000, 24).astype(np.float16), _unit(rng, 30000, 24).astype(np.float16)
    DB[20000] = DB[7]
    idx, score = matching.nn_search(torch.from_numpy(Q)[None].to(dev), torch.from_numpy(DB)[None].to(dev), return_score=True)
    idx, score = idx[0].cpu().numpy(), score[0].cpu().numpy()
    ref, best, second = om.nn_search(Q.astype(np.float64), DB.astype(np.float64))
    clear = best - second > 4 * EPS
    assert clear.mean() > 0.9 and np.array_equal(idx[clear], ref[clear])
    chosen = np.einsum("sd,sd->s", Q.astype(np.float64), DB[idx].astype(np.float64))
    assert np.all(chosen >= best - 4 * EPS) and np.abs(score - chosen).max() < 4 * EPS
    assert np.all(idx[(ref == 7) & clear] == 7)


def test_nn_search_batched_and_rejects_bad_input(dev):
    rng = np.random.default_rng(3)
    Q = np.stack([_unit(rng, 100, 24) for _ in range(3)]); DB = np.stack([_unit(rng, 900, 24) for _ in range(3)])
    idx = matching.nn_search(torch.from_numpy(Q).to(dev), torch.from_numpy(DB).to(dev)).cpu().numpy()
    for b in range(3):
        ref, best, second = om.nn_search(Q[b], DB[b])
        ok = best - second > 4 * EPS
        assert np.array_equal(idx[b][ok], ref[ok])
    with pytest.raises(RuntimeError):
        matching.nn_search(torch.from_numpy(Q), torch.from_numpy(DB))           # CPU tensors
    with pytest.raises(RuntimeError, match="invalid argument"):
        matching.nn_search(torch.zeros(1, 4, 20, device=dev), torch.zeros(1, 9, 20, device=dev))   # unsupported D


def test_fast_reciprocal_nn_on_a_smooth_scene(dev):
    """Two views of a smooth surface with position-encoding descriptors: reciprocal matches must equal the oracle's
    and land on the true correspondence."""
    sc = synthetic.geometric_pair(64, 96, seed=4, batch=1)
    D1, D2 = sc["D21"][0], sc["D11"][0]                                        # seeds in view 2's pixels, search in view 1
    i1, i2 = matching.fast_reciprocal_nn(torch.from_numpy(D1).to(dev), torch.from_numpy(D2).to(dev), subsample=4)
    r1, r2 = om.fast_reciprocal_nn(D1, D2, subsample=4)
    got = set(zip(i1.cpu().tolist(), i2.cpu().tolist())); ref = set(zip(r1.tolist(), r2.tolist()))
    assert len(ref) > 100 and len(got & ref) >= 0.98 * len(ref) and len(got - ref) <= 0.02 * len(ref)
    # the sync-free device loop returns the same set, also from fp16 descriptors
    j1, j2 = matching.fast_reciprocal_nn_device(torch.from_numpy(D1).to(dev), torch.from_numpy(D2).to(dev), subsample=4)
    assert set(zip(j1.cpu().tolist(), j2.cpu().tolist())) == got
    k1, k2 = matching.fast_reciprocal_nn_device(torch.from_numpy(D1).to(dev).half(), torch.from_numpy(D2).to(dev).half(), subsample=4)
    gk = set(zip(k1.cpu().tolist(), k2.cpu().tolist()))
    assert len(gk & ref) >= 0.95 * len(ref)
    uv = sc["uv_true"][0][i1.cpu().numpy()]                                    # where each matched view-2 pixel truly lands
    x2, y2 = i2.cpu().numpy() % 96, i2.cpu().numpy() // 96
    inside = (uv[:, 0] > 1) & (uv[:, 0] < 94) & (uv[:, 1] > 1) & (uv[:, 1] < 62)
    err = np.hypot(x2 - uv[:, 0], y2 - uv[:, 1])[inside]
    assert np.median(err) < 1.0 and np.percentile(err, 95) < 2.5


def test_fast_reciprocal_nn_batched_equals_per_pair(dev):
    """P pairs in one set of launches (m3_frnn_pack once per map, m3_frnn_round per round) give, pair by pair, exactly
    the single-pair result - fp32 and fp16 descriptors; pairs of different content, so a mix-up of the pair axis shows."""
    scs = [synthetic.geometric_pair(64, 96, seed=10 + k, batch=1) for k in range(3)]
    D1 = torch.from_numpy(np.stack([s["D21"][0] for s in scs])).to(dev)
    D2 = torch.from_numpy(np.stack([s["D11"][0] for s in scs])).to(dev)
    for half in (False, True):
        A, B = (D1.half(), D2.half()) if half else (D1, D2)
        pid, i1, i2 = matching.fast_reciprocal_nn_device(A, B, subsample=4, max_iter=6)
        assert pid.dtype == torch.int64 and bool((pid[1:] >= pid[:-1]).all())
        for k in range(3):
            s1, s2 = matching.fast_reciprocal_nn_device(A[k], B[k], subsample=4, max_iter=6)
            m = pid == k
            assert torch.equal(i1[m], s1) and torch.equal(i2[m], s2) and s1.numel() > 100
    with pytest.raises(ValueError):
        matching.fast_reciprocal_nn_device(D1, D2[:2])


def test_fast_reciprocal_nn_maps_fixed_shapes_graph_capture_and_active_rounds(dev):
    """fast_reciprocal_nn_maps: the whole matcher on the device with fixed-shape outputs (no torch.unique, no host
    synchronisation) - (1) its sorted pair list equals the set a sort-based reference builds from full rounds
    (m3_frnn_round on every seed in every round: the active-set rounds must not change a result), (2) the tracker maps
    are the same pairs seen from view 2, (3) it can be captured into a hipGraph and replayed on new inputs."""
    from mast3r_slam import _ffi
    scs = [synthetic.geometric_pair(64, 96, seed=30 + k, batch=1) for k in range(3)]
    D1 = torch.from_numpy(np.stack([s["D21"][0] for s in scs])).to(dev).half()
    D2 = torch.from_numpy(np.stack([s["D11"][0] for s in scs])).to(dev).half()
    P, h, w, d = D1.shape
    n = h * w
    rounds, sub = 5, 4
    m = matching.fast_reciprocal_nn_maps(D1, D2, subsample=sub, max_iter=rounds)
    counts = m["count"].cpu().tolist()
    s = m["seeds"]
    assert m["pairs"].shape == (P, s, 2) and m["idx"].shape == (P, n) and m["valid"].shape == (P, n, 1)
    # reference: full rounds through the C ABI + a host-side sort / unique
    L = _ffi.lib()
    st = _ffi.stream_ptr()
    pk1 = torch.empty(int(L.m3_frnn_pack_bytes(P, n, 1)), dtype=torch.uint8, device=dev)
    pk2 = torch.empty_like(pk1)
    _ffi.call("m3_frnn_pack", _ffi.ptr(D1), _ffi.ptr(pk1), P, n, d, 1, st)
    _ffi.call("m3_frnn_pack", _ffi.ptr(D2), _ffi.ptr(pk2), P, n, d, 1, st)
    ys, xs = torch.arange(sub // 2, h, sub, device=dev), torch.arange(sub // 2, w, sub, device=dev)
    cur = (ys[:, None] * w + xs[None, :]).reshape(-1).to(torch.int32)[None].repeat(P, 1).contiguous()
    active = torch.ones((P, s), dtype=torch.uint8, device=dev)
    got1 = torch.empty((rounds, P, s), dtype=torch.int32, device=dev); got2 = torch.empty_like(got1)
    xy2 = torch.empty((P, s), dtype=torch.int32, device=dev); keys = torch.zeros((P, s), dtype=torch.int64, device=dev)
    for r in range(rounds):
        _ffi.call("m3_frnn_round", _ffi.ptr(pk1), _ffi.ptr(pk2), _ffi.ptr(cur), _ffi.ptr(active), _ffi.ptr(got1[r]), _ffi.ptr(got2[r]),
                  _ffi.ptr(xy2), _ffi.ptr(keys), P, s, n, n, 1, st)
    later = int((got1[1:] >= 0).sum())
    assert later > 0                                                 # some seeds do converge after round 0: the active rounds matter
    for p in range(P):
        keep = got1[:, p].reshape(-1) >= 0
        ref = torch.unique(torch.stack([got1[:, p].reshape(-1)[keep], got2[:, p].reshape(-1)[keep]], 1), dim=0)
        assert counts[p] == ref.shape[0] and counts[p] > 100
        assert torch.equal(m["pairs"][p, :counts[p]], ref)           # sorted by p1, as torch.unique sorts
        assert bool((m["pairs"][p, counts[p]:] == -1).all())
        v = m["valid"][p, :, 0]
        assert int(v.sum()) == counts[p]
        assert torch.equal(m["idx"][p][ref[:, 1].long()], ref[:, 0].long())
        assert torch.equal(m["map1"][p][ref[:, 0].long()], ref[:, 1]) and int((m["map1"][p] >= 0).sum()) == counts[p]
    # hipGraph capture: static inputs, replay on other data
    a, b = D1.clone(), D2.clone()
    for _ in range(2):
        matching.fast_reciprocal_nn_maps(a, b, subsample=sub, max_iter=rounds)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = matching.fast_reciprocal_nn_maps(a, b, subsample=sub, max_iter=rounds)
    a.copy_(D1.flip(0)); b.copy_(D2.flip(0))
    g.replay()
    torch.cuda.synchronize()
    assert out["count"].cpu().tolist() == counts[::-1]
    assert torch.equal(out["pairs"], m["pairs"].flip(0)) and torch.equal(out["idx"], m["idx"].flip(0))


def _same_maps(a, b):
    for k in ("map1", "pairs", "count", "idx", "valid"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("case", ["smooth_f32", "smooth_f16", "random_f32", "random_f16", "ties_f16", "ragged_f32", "ragged_f16"])
def test_block_bound_search_equals_brute_force_bit_for_bit(dev, case):
    """m3_frnn_round_pruned (block bounds: centroid search -> lower bound -> survivor bits -> exact scoring of the
    surviving 64-row blocks with the brute-force MFMA sequence) against m3_frnn_round / m3_frnn_round_active on
      * smooth scenes (the case it is built for: a few blocks per query tile survive),
      * random unit descriptors (no spatial coherence: nearly every block survives, the device-side gate hands the search
        to the brute-force kernel),
      * a map tiled from a small patch (every descriptor occurs many times: ties must go to the LOWEST index, so blocks
        that merely tie with the maximum must survive too),
      * maps whose row count is not a multiple of the 64-row block (the last block is short) nor of 16.
    Every output of fast_reciprocal_nn_maps is compared with torch.equal."""
    kind, dt = case.split("_")
    half = dt == "f16"
    rng = np.random.default_rng(7)
    if kind == "smooth":
        scs = [synthetic.geometric_pair(64, 96, seed=50 + k, batch=1) for k in range(3)]
        D1 = np.stack([s["D21"][0] for s in scs]); D2 = np.stack([s["D11"][0] for s in scs])
    elif kind == "random":
        D1 = rng.normal(size=(2, 48, 64, 24)).astype(np.float32); D2 = rng.normal(size=(2, 48, 64, 24)).astype(np.float32)
        D1 /= np.linalg.norm(D1, axis=-1, keepdims=True); D2 /= np.linalg.norm(D2, axis=-1, keepdims=True)
    elif kind == "ties":
        patch = rng.normal(size=(1, 8, 16, 24)).astype(np.float32)
        patch /= np.linalg.norm(patch, axis=-1, keepdims=True)
        D1 = np.tile(patch, (2, 6, 6, 1)); D2 = np.tile(patch[:, ::-1], (2, 6, 6, 1)).copy()
    else:
        sc = synthetic.geometric_pair(50, 70, seed=61, batch=2)                 # 3500 rows: 54 blocks + 44 rows
        D1, D2 = sc["D21"], sc["D11"]
    A, B = torch.from_numpy(np.ascontiguousarray(D1)).to(dev), torch.from_numpy(np.ascontiguousarray(D2)).to(dev)
    if half:
        A, B = A.half(), B.half()
    for sub, rounds in ((4, 4), (8, 2)):
        fast = matching.fast_reciprocal_nn_maps(A, B, subsample=sub, max_iter=rounds, prune=True)
        brute = matching.fast_reciprocal_nn_maps(A, B, subsample=sub, max_iter=rounds, prune=False)
        _same_maps(fast, brute)
        if kind in ("smooth", "ragged"):
            assert int(brute["count"].min()) > 20


def test_match_dispatches_to_the_fast_reciprocal_nn_matcher(dev):
    """matching.match with matching.use_fast_nn (the switch is this repo's; the contract is matching.py:12-38's): idx / valid
    in the dense matchers' format - valid exactly where a reciprocal pair ends AND the 3-D points agree within dist_thresh;
    on the synthetic two-view scene most seeds match and the matches are the true correspondences; the FrameTracker's
    gather + GN accept the maps as they are."""
    from mast3r_slam import config
    sc = synthetic.geometric_pair(64, 96, seed=77, batch=2)
    X11, X21 = torch.from_numpy(sc["X11"]).to(dev), torch.from_numpy(sc["X21"]).to(dev)
    D11, D21 = torch.from_numpy(sc["D11"]).to(dev), torch.from_numpy(sc["D21"]).to(dev)
    config.set_config({"matching": {"use_fast_nn": True, "fast_nn_subsample": 4, "fast_nn_rounds": 4}})
    try:
        idx, valid = matching.match(X11, X21, D11, D21)
    finally:
        config.set_config({"matching": {"use_fast_nn": False, "fast_nn_subsample": 8, "fast_nn_rounds": 3}})
    b, h, w, _ = X11.shape
    n = h * w
    assert idx.shape == (b, n) and idx.dtype == torch.int64 and valid.shape == (b, n, 1) and valid.dtype == torch.bool
    m = matching.fast_reciprocal_nn_maps(D11, D21, subsample=4, max_iter=4)
    near = (X11.reshape(b, n, 3).gather(1, m["idx"][:, :, None].expand(b, n, 3)) - X21.reshape(b, n, 3)).norm(dim=-1) < 0.1
    assert torch.equal(valid[:, :, 0], m["valid"][:, :, 0] & near)
    assert torch.equal(idx[valid[:, :, 0]], m["idx"][valid[:, :, 0]])
    seeds = (h // 4) * (w // 4)
    assert int(valid.sum()) > 0.7 * b * seeds
    # matched view-2 pixels land on their true view-1 positions (uv_true: where each view-2 pixel lies in view 1)
    for p in range(b):
        v = valid[p, :, 0].cpu().numpy()
        got = idx[p].cpu().numpy()[v]
        uv = sc["uv_true"][p][v]
        err = np.hypot(got % w - uv[:, 0], got // w - uv[:, 1])
        assert np.median(err) < 1.0 and np.percentile(err, 95) < 2.5
