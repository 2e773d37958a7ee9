"""Dense matching front end on the MI355X.

Mirror of /root/reference/src/mlx_mast3r_slam/matching.py (match :12, match_simple :41,
pixel_to_lin :93, lin_to_pixel :106, prep_for_iter_proj :134, match_iterative_proj :339)
over torch tensors on the ROCm device.  Every function is batched over B pairs and
stream-ordered; nothing synchronises with the host.
"""
from __future__ import annotations

import torch

from . import _ffi
from .config import get_config
from . import kernels


def pixel_to_lin(p: torch.Tensor, w: int) -> torch.Tensor:
    """matching.py:93-103."""
    return p[..., 0] + w * p[..., 1]


def lin_to_pixel(idx: torch.Tensor, w: int) -> torch.Tensor:
    """matching.py:106-118."""
    return torch.stack([idx % w, idx // w], dim=-1)


def _check_maps(X11, X21):
    X11 = _ffi.check(X11, torch.float32, "X11")
    if X11.dim() != 4 or X11.shape[-1] != 3:
        raise ValueError(f"X11 must be [B,H,W,3], got {tuple(X11.shape)}")
    b, h, w, _ = X11.shape
    X21 = _ffi.check(X21, torch.float32, "X21")
    if X21.numel() != b * h * w * 3:
        raise ValueError(f"X21 must hold {b}x{h}x{w}x3 values, got {tuple(X21.shape)}")
    return X11, X21, b, h, w


def _check_idx(idx, b, n):
    if idx is None:
        return None
    idx = _ffi.check(idx, torch.int64, "idx_1_to_2_init") if idx.dtype == torch.int64 else \
        _ffi.check(idx.to(torch.int64), torch.int64, "idx_1_to_2_init")
    if tuple(idx.shape) != (b, n):
        raise ValueError(f"idx_1_to_2_init must be [{b},{n}], got {tuple(idx.shape)}")
    return idx


def prep_for_iter_proj(X11, X21, idx_1_to_2_init=None):
    """matching.py:134-175 -> (rays_with_grad [B,H,W,9], pts3d_norm [B,N,3], p_init [B,N,2])."""
    X11, X21, b, h, w = _check_maps(X11, X21)
    n = h * w
    idx = _check_idx(idx_1_to_2_init, b, n)
    dev = X11.device
    rays = torch.empty((b, h, w, 9), dtype=torch.float32, device=dev)
    tgt = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
    p0 = torch.empty((b, n, 2), dtype=torch.float32, device=dev)
    _ffi.call("m3_prep_iter_proj", _ffi.ptr(X11), _ffi.ptr(X21), _ffi.ptr(idx), _ffi.ptr(rays), _ffi.ptr(tgt),
              _ffi.ptr(p0), b, h, w, _ffi.stream_ptr())
    return rays, tgt, p0


def match_simple(X11, X21, D11=None, D21=None, idx_1_to_2_init=None):
    """matching.py:41-90 -> (idx [B,N] int64, valid [B,N,1] bool)."""
    cfg = get_config()["matching"]
    X11, X21, b, h, w = _check_maps(X11, X21)
    n = h * w
    idx = _check_idx(idx_1_to_2_init, b, n)
    out = torch.empty((b, n), dtype=torch.int64, device=X11.device)
    valid = torch.empty((b, n), dtype=torch.uint8, device=X11.device)
    _ffi.call("m3_match_simple", _ffi.ptr(X11), _ffi.ptr(X21), _ffi.ptr(idx), _ffi.ptr(out), _ffi.ptr(valid),
              b, h, w, float(cfg["dist_thresh"]), _ffi.stream_ptr())
    return out, valid.bool()[:, :, None]


def match_iterative_proj(X11, X21, D11, D21, idx_1_to_2_init=None, *, stop_scope: str = "batch"):
    """matching.py:339-461: prep -> iter_proj -> refine_matches -> occlusion test -> linear index.

    Returns (idx_1_to_2 [B,N] int64, valid [B,N,1] bool).  stop_scope="batch" makes a batched
    call equal to B separate calls (the reference only ever calls this with B=1)."""
    cfg = get_config()["matching"]
    X11, X21, b, h, w = _check_maps(X11, X21)
    n = h * w
    rays, tgt, p0 = prep_for_iter_proj(X11, X21, idx_1_to_2_init)
    p1, valid_proj = kernels.iter_proj(rays, tgt, p0, cfg["max_iter"], cfg["lambda_init"],
                                       cfg["convergence_thresh"], stop_scope=stop_scope)
    dev = X11.device
    vp = valid_proj.to(torch.uint8)
    p_int = None
    if cfg.get("use_refine", True) and cfg.get("refine_radius", 3) > 0:
        D11 = _ffi.check(D11, (torch.float32, torch.float16), "D11")       # float16 = "fp16 features"
        D21 = _ffi.check(D21, D11.dtype, "D21")
        d = D11.shape[-1]
        p_trunc = torch.empty((b, n, 2), dtype=torch.int32, device=dev)
        _ffi.call("m3_trunc_i32", _ffi.ptr(p1), _ffi.ptr(p_trunc), b * n * 2, _ffi.stream_ptr())
        p_int = kernels.refine_matches(D11.reshape(b, h, w, d), D21.reshape(b, n, d), p_trunc,
                                       cfg.get("refine_radius", 3), cfg.get("refine_dilation", 2),
                                       chained=cfg.get("refine_chained", False))
    idx = torch.empty((b, n), dtype=torch.int64, device=dev)
    valid = torch.empty((b, n), dtype=torch.uint8, device=dev)
    _ffi.call("m3_match_epilogue", _ffi.ptr(X11), _ffi.ptr(X21), _ffi.ptr(p_int), _ffi.ptr(p1), _ffi.ptr(vp),
              _ffi.ptr(idx), _ffi.ptr(valid), b, h, w, float(cfg["dist_thresh"]), _ffi.stream_ptr())
    return idx, valid.bool()[:, :, None]


def match(X11, X21, D11, D21, idx_1_to_2_init=None):
    """matching.py:12-38: dispatch on config matching.use_simple (default True); matching.use_fast_nn (not in the reference,
    default False) selects the fast reciprocal NN matcher instead."""
    if get_config().get("matching", {}).get("use_fast_nn", False):
        return match_fast_nn(X11, X21, D11, D21, idx_1_to_2_init)
    if get_config().get("matching", {}).get("use_simple", True):
        return match_simple(X11, X21, D11, D21, idx_1_to_2_init)
    return match_iterative_proj(X11, X21, D11, D21, idx_1_to_2_init)


# ---------------------------------------------------------------------------------------------------
# Fast reciprocal nearest-neighbour matching (named by BASELINE.json; MASt3R, Leroy et al. 2024, sec. 3.3).
# The reference tree has no implementation: semantics and oracle are this repo's.
def nn_search(Q: torch.Tensor, DB: torch.Tensor, return_score: bool = False, method: str = "mfma"):
    """Q [B,S,D], DB [B,N,D] -> idx int32 [B,S] = argmax_n <Q[b,s], DB[b,n]>, ties to the lowest n.
    method "mfma" (default): m3_nn_search_mfma - the scores are a GEMM on v_mfma_f32_16x16x32_f16; fp16 inputs
    (descriptors stored as fp16) take one MFMA per 16 x 16 scores with exact products, fp32 inputs are split hi + lo
    (3 MFMAs, error <= 2^-24 for unit vectors).  method "fma": m3_nn_search, the fp32 FMA-chain kernel (fp32 only)."""
    if method not in ("mfma", "fma"):
        raise ValueError("method must be 'mfma' or 'fma'")
    dts = (torch.float32, torch.float16) if method == "mfma" else torch.float32
    Q = _ffi.check(Q, dts, "Q")
    DB = _ffi.check(DB, dts, "DB")
    if Q.dtype != DB.dtype:
        raise TypeError(f"Q and DB must have the same dtype, got {Q.dtype} / {DB.dtype}")
    if Q.dim() != 3 or DB.dim() != 3 or Q.shape[0] != DB.shape[0] or Q.shape[2] != DB.shape[2]:
        raise ValueError(f"Q [B,S,D] / DB [B,N,D] expected, got {tuple(Q.shape)} / {tuple(DB.shape)}")
    b, s, d = Q.shape
    n = DB.shape[1]
    idx = torch.empty((b, s), dtype=torch.int32, device=Q.device)
    score = torch.empty((b, s), dtype=torch.float32, device=Q.device) if return_score else None
    keys = torch.empty((b, s), dtype=torch.int64, device=Q.device)
    if method == "fma":
        _ffi.call("m3_nn_search", _ffi.ptr(Q), _ffi.ptr(DB), _ffi.ptr(idx), _ffi.ptr(score), _ffi.ptr(keys), b, s, n, d,
                  _ffi.stream_ptr())
    else:
        f16 = 1 if Q.dtype == torch.float16 else 0
        ws = torch.empty(int(_ffi.lib().m3_nn_pack_bytes(b, s, n, f16)), dtype=torch.uint8, device=Q.device)
        _ffi.call("m3_nn_search_mfma", _ffi.ptr(Q), _ffi.ptr(DB), _ffi.ptr(idx), _ffi.ptr(score), _ffi.ptr(keys), _ffi.ptr(ws),
                  b, s, n, d, f16, _ffi.stream_ptr())
    return (idx, score) if return_score else idx


_PATCH_ORDER: dict = {}


def _seed_patch_order(hs: int, ws: int, dev) -> torch.Tensor:
    """Seed slots (row-major hs x ws grid) listed patch by patch: 4 x 4 seeds = one 16-query tile of the pruned search, four
    consecutive tiles = an 8 x 8 patch.  The queries of a tile / group then land close together in the other view and need
    few blocks between them (m3_frnn_round_pruned's seed_order; results do not depend on it)."""
    key = (hs, ws, str(dev))
    if key not in _PATCH_ORDER:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # argsort allocates and the table is cached across calls: it must exist before a capture starts
            raise RuntimeError("fast_reciprocal_nn_maps: call once eagerly (warm-up) for this map size before capturing it into a graph")
        iy, ix = torch.meshgrid(torch.arange(hs, device=dev), torch.arange(ws, device=dev), indexing="ij")    # no host -> device copy
        iy, ix = iy.reshape(-1), ix.reshape(-1)
        k = ((iy // 8) * ((ws + 7) // 8) + ix // 8) * 64 + (((iy % 8) // 4) * 2 + (ix % 8) // 4) * 16 + (iy % 4) * 4 + ix % 4
        _PATCH_ORDER[key] = torch.argsort(k, stable=True).to(torch.int32)
    return _PATCH_ORDER[key]


def fast_reciprocal_nn_maps(D1: torch.Tensor, D2: torch.Tensor, subsample: int = 8, max_iter: int = 10, tracker_maps: bool = True,
                            prune: bool = True):
    """Fast reciprocal NN for a batch of P pairs (D1, D2 [P,H,W,D], fp32 or fp16) with EVERYTHING on the device and every
    output of a fixed shape - no host synchronisation, no data-dependent allocation - so the call can be captured into a
    hipGraph.  Each descriptor map is packed to K-padded fp16 once (m3_frnn_pack) and serves as the database of one
    search direction and the query source of the other; round 0 runs on all seeds (m3_frnn_round), later rounds only on
    the seeds that have not converged yet (m3_frnn_round_active: a device-side ordered compaction feeds the searches);
    m3_frnn_collect turns the rounds' reciprocal pairs into
        map1   int32 [P,N1]    view-1 pixel -> its reciprocal partner in view 2 (-1 = none)
        idx    int64 [P,N2], valid bool [P,N2,1]   the tracker's maps (view-2 pixel -> view-1 pixel)   (tracker_maps)
        pairs  int32 [P,S,2], count int32 [P]      the distinct (p1, p2) per pair, sorted by p1; rows >= count are -1.
    prune (default): the searches use block bounds (m3_frnn_blockstats once per map, m3_frnn_round_pruned) - the same
    arg-max bit for bit, most of the other view never scored when the descriptor maps are spatially coherent, the
    brute-force kernel as the device-side fallback when they are not.  prune=False: brute force always."""
    if D1.dim() != 4 or D2.dim() != 4 or D1.shape[-1] != D2.shape[-1] or D1.shape[0] != D2.shape[0]:
        raise ValueError("D1, D2 must be [P,H,W,D] with the same P and D")
    if D1.dtype != D2.dtype:
        raise TypeError(f"D1 and D2 must have the same dtype, got {D1.dtype} / {D2.dtype}")
    A = _ffi.check(D1, (torch.float32, torch.float16), "D1")
    B = _ffi.check(D2, A.dtype, "D2")
    P, h1, w1, d = A.shape
    n1, n2 = h1 * w1, B.shape[1] * B.shape[2]
    dev = A.device
    ys = torch.arange(subsample // 2, h1, subsample, device=dev)
    xs = torch.arange(subsample // 2, w1, subsample, device=dev)
    seeds = (ys[:, None] * w1 + xs[None, :]).reshape(-1).to(torch.int32)
    s = seeds.numel()
    if s == 0 or max_iter <= 0:
        raise ValueError("fast_reciprocal_nn_maps needs at least one seed and one round")
    L = _ffi.lib()
    f16 = 1 if A.dtype == torch.float16 else 0
    st = _ffi.stream_ptr()
    pk1 = torch.empty(int(L.m3_frnn_pack_bytes(P, n1, f16)), dtype=torch.uint8, device=dev)
    pk2 = torch.empty(int(L.m3_frnn_pack_bytes(P, n2, f16)), dtype=torch.uint8, device=dev)
    _ffi.call("m3_frnn_pack", _ffi.ptr(A), _ffi.ptr(pk1), P, n1, d, f16, st)
    _ffi.call("m3_frnn_pack", _ffi.ptr(B), _ffi.ptr(pk2), P, n2, d, f16, st)
    cur = seeds[None].repeat(P, 1).contiguous()
    active = torch.ones((P, s), dtype=torch.uint8, device=dev)
    got1 = torch.empty((max_iter, P, s), dtype=torch.int32, device=dev)
    got2 = torch.empty((max_iter, P, s), dtype=torch.int32, device=dev)
    xy2 = torch.empty((P, s), dtype=torch.int32, device=dev)
    keys = torch.zeros((P, s), dtype=torch.int64, device=dev)
    act_ws = torch.empty(P * (s + 1), dtype=torch.int32, device=dev)
    if prune:
        h2, w2 = B.shape[1], B.shape[2]
        st1 = torch.empty(int(L.m3_frnn_stats_bytes(P, h1, w1)), dtype=torch.uint8, device=dev)
        st2 = torch.empty(int(L.m3_frnn_stats_bytes(P, h2, w2)), dtype=torch.uint8, device=dev)
        pws = torch.empty(int(L.m3_frnn_prune_ws_bytes(P, s, h1, w1, h2, w2)), dtype=torch.uint8, device=dev)
        _ffi.call("m3_frnn_blockstats", _ffi.ptr(pk1), _ffi.ptr(st1), P, h1, w1, f16, st)
        _ffi.call("m3_frnn_blockstats", _ffi.ptr(pk2), _ffi.ptr(st2), P, h2, w2, f16, st)
        order = _seed_patch_order(len(ys), len(xs), dev)
    for r in range(max_iter):
        if prune:
            _ffi.call("m3_frnn_round_pruned", _ffi.ptr(pk1), _ffi.ptr(pk2), _ffi.ptr(st1), _ffi.ptr(st2), _ffi.ptr(cur),
                      _ffi.ptr(active), _ffi.ptr(got1[r]), _ffi.ptr(got2[r]), _ffi.ptr(xy2), _ffi.ptr(keys),
                      _ffi.ptr(act_ws), _ffi.ptr(order), _ffi.ptr(pws), P, s, h1, w1, h2, w2, f16, st)
        elif r == 0:
            _ffi.call("m3_frnn_round", _ffi.ptr(pk1), _ffi.ptr(pk2), _ffi.ptr(cur), _ffi.ptr(active), _ffi.ptr(got1[r]),
                      _ffi.ptr(got2[r]), _ffi.ptr(xy2), _ffi.ptr(keys), P, s, n1, n2, f16, st)
        else:
            _ffi.call("m3_frnn_round_active", _ffi.ptr(pk1), _ffi.ptr(pk2), _ffi.ptr(cur), _ffi.ptr(active), _ffi.ptr(got1[r]),
                      _ffi.ptr(got2[r]), _ffi.ptr(xy2), _ffi.ptr(keys), _ffi.ptr(act_ws), P, s, n1, n2, f16, st)
    map1 = torch.empty((P, n1), dtype=torch.int32, device=dev)
    idx = torch.empty((P, n2), dtype=torch.int64, device=dev) if tracker_maps else None
    valid = torch.empty((P, n2), dtype=torch.uint8, device=dev) if tracker_maps else None
    pairs = torch.empty((P, s, 2), dtype=torch.int32, device=dev)
    count = torch.empty((P,), dtype=torch.int32, device=dev)
    chunk_ws = torch.empty(P * int(L.m3_frnn_chunks(n1)), dtype=torch.int32, device=dev)
    _ffi.call("m3_frnn_collect", _ffi.ptr(got1), _ffi.ptr(got2), max_iter, P, s, n1, n2, _ffi.ptr(map1), _ffi.ptr(idx),
              _ffi.ptr(valid), _ffi.ptr(pairs), _ffi.ptr(count), _ffi.ptr(chunk_ws), st)
    out = dict(map1=map1, pairs=pairs, count=count, seeds=s)
    if tracker_maps:
        out["idx"], out["valid"] = idx, valid.view(torch.bool)[:, :, None]
    return out


def use_fast_nn_enabled() -> bool:
    return bool(get_config().get("matching", {}).get("use_fast_nn", False))


def match_fraction_scale(h: int, w: int) -> float:
    """What a "fraction of the pixels that matched" has to be multiplied with so that the reference's thresholds
    (tracking.min_match_frac, match_frac_thresh, the factor graph's min_match_frac) keep their meaning: 1 for the dense
    matchers; with matching.use_fast_nn the matcher yields at most one match per SEED, so pixels / seeds."""
    cfg = get_config().get("matching", {})
    if not cfg.get("use_fast_nn", False):
        return 1.0
    sub = int(cfg.get("fast_nn_subsample", 8))
    seeds = len(range(sub // 2, h, sub)) * len(range(sub // 2, w, sub))
    return float(h * w) / float(max(1, seeds))


def match_fast_nn(X11, X21, D11, D21, idx_1_to_2_init=None):
    """The matcher BASELINE.json's north_star names, behind matching.match's contract (matching.py:12-38): descriptors
    D11 [B,H,W,D] (the view the indices point INTO) and D21 [B,H,W,D] or [B,N,D] -> (idx_1_to_2 [B,N] int64, valid [B,N,1]
    bool).  A pixel n of the second map is matched when it is the view-2 end of a reciprocal nearest-neighbour pair
    (fast_reciprocal_nn_maps: seeds on a grid in view 1, `fast_nn_rounds` rounds) AND its 3-D points agree:
    |X11[idx[n]] - X21[n]| < dist_thresh, the occlusion test of the reference's matchers (matching.py:75-90, :438-452).
    The result is sparse by construction (at most one match per seed); idx_1_to_2_init is accepted and ignored (the seeds
    do not depend on a previous match).  Everything stays on the stream: no host synchronisation."""
    cfg = get_config()["matching"]
    X11, X21, b, h, w = _check_maps(X11, X21)
    n = h * w
    D11 = _ffi.check(D11, (torch.float32, torch.float16), "D11")
    D21 = _ffi.check(D21, D11.dtype, "D21")
    d = D11.shape[-1]
    if D11.numel() != b * n * d or D21.numel() != b * n * d:
        raise ValueError(f"D11 / D21 must hold {b}x{h}x{w}x{d} values, got {tuple(D11.shape)} / {tuple(D21.shape)}")
    m = fast_reciprocal_nn_maps(D11.reshape(b, h, w, d), D21.reshape(b, h, w, d), subsample=int(cfg.get("fast_nn_subsample", 8)),
                                max_iter=int(cfg.get("fast_nn_rounds", 3)), tracker_maps=True)
    idx = torch.empty((b, n), dtype=torch.int64, device=X11.device)
    near = torch.empty((b, n), dtype=torch.uint8, device=X11.device)
    _ffi.call("m3_match_simple", _ffi.ptr(X11), _ffi.ptr(X21), _ffi.ptr(m["idx"]), _ffi.ptr(idx), _ffi.ptr(near),
              b, h, w, float(cfg["dist_thresh"]), _ffi.stream_ptr())
    return idx, (m["valid"][:, :, 0] & near.bool())[:, :, None]


def fast_reciprocal_nn_device(D1: torch.Tensor, D2: torch.Tensor, subsample: int = 8, max_iter: int = 10):
    """fast_reciprocal_nn with the whole loop on the device, for ONE pair (D1, D2 [H,W,D]) or a BATCH of P pairs
    ([P,H,W,D] each) in the same launches (fast_reciprocal_nn_maps).  Same result set as the shrinking-active-set version
    (fast_reciprocal_nn) for the same max_iter.  fp32 or fp16 descriptors.  ONE host synchronisation, at the end: the pair
    counts, to cut the fixed-capacity device list to size.
    Returns (idx1, idx2) int64 [M] for one pair; (pair, idx1, idx2) int64 [M] (sorted by pair, idx1) for a batch."""
    batched = D1.dim() == 4
    if D1.dim() not in (3, 4) or D2.dim() != D1.dim() or D1.shape[-1] != D2.shape[-1] or (batched and D1.shape[0] != D2.shape[0]):
        raise ValueError("D1, D2 must be [H,W,D] or [P,H,W,D] with the same D (and P)")
    if D1.dtype != D2.dtype:
        raise TypeError(f"D1 and D2 must have the same dtype, got {D1.dtype} / {D2.dtype}")
    A, B = (D1, D2) if batched else (D1[None], D2[None])
    dev = A.device
    e = torch.empty(0, dtype=torch.int64, device=dev)
    if max_iter <= 0 or A.shape[1] <= subsample // 2 or A.shape[2] <= subsample // 2:
        return (e, e, e) if batched else (e, e)
    m = fast_reciprocal_nn_maps(A, B, subsample, max_iter, tracker_maps=False)
    counts = m["count"].cpu().tolist()                                    # the one host synchronisation
    rows = [m["pairs"][p, :c].long() for p, c in enumerate(counts)]
    pid = torch.cat([torch.full((c,), p, dtype=torch.int64, device=dev) for p, c in enumerate(counts)]) if counts else e
    trip = torch.cat(rows) if rows else torch.empty((0, 2), dtype=torch.int64, device=dev)
    return (pid, trip[:, 0], trip[:, 1]) if batched else (trip[:, 0], trip[:, 1])


def fast_reciprocal_nn(D1: torch.Tensor, D2: torch.Tensor, subsample: int = 8, max_iter: int = 10):
    """Reciprocal matches between two descriptor maps D1, D2 [H,W,D] (L2-normalised, f32).

    Seeds = view-1 pixels on a grid of step `subsample` (offset subsample//2).  Each round maps the active
    view-1 pixels to their nearest neighbour in view 2 and back; a seed has CONVERGED when the round
    returns to the pixel it started from (a mutual nearest neighbour pair), otherwise it continues from
    where it landed, for at most max_iter rounds.  Returns (idx1, idx2) int64 [M]: linear pixel indices of
    the distinct converged pairs, sorted by idx1.  One host sync per round (the active set shrinks)."""
    if D1.dim() != 3 or D2.dim() != 3 or D1.shape[2] != D2.shape[2]:
        raise ValueError("D1, D2 must be [H,W,D] with the same D")
    h1, w1, d = D1.shape
    f1, f2 = D1.reshape(1, -1, d).contiguous(), D2.reshape(1, -1, d).contiguous()
    dev = D1.device
    ys = torch.arange(subsample // 2, h1, subsample, device=dev)
    xs = torch.arange(subsample // 2, w1, subsample, device=dev)
    xy1 = (ys[:, None] * w1 + xs[None, :]).reshape(-1)                      # active view-1 pixels
    out1, out2 = [], []
    for _ in range(max_iter):
        if xy1.numel() == 0:
            break
        xy2 = nn_search(f1[:, xy1], f2)[0].long()                           # view 1 -> view 2
        back = nn_search(f2[:, xy2], f1)[0].long()                          # view 2 -> view 1
        conv = back == xy1
        out1.append(xy1[conv]); out2.append(xy2[conv])
        xy1 = torch.unique(back[~conv])                                     # continue from where they landed
    if not out1:
        e = torch.empty(0, dtype=torch.int64, device=dev)
        return e, e
    pairs = torch.unique(torch.stack([torch.cat(out1), torch.cat(out2)], 1), dim=0)
    return pairs[:, 0], pairs[:, 1]
