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
