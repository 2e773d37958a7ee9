"""Frame-to-keyframe tracking on the MI355X.

Mirror of /root/reference/src/mlx_mast3r_slam/tracker.py: FrameTracker.track :51-175,
_get_points_poses :177-214, _solve :216-256, _opt_pose_ray_dist_sim3 :258-324.  The
Gauss-Newton loop (residuals, Huber, J^T J / J^T r reduction, 7x7 solve, Sim3 retraction,
stop test) runs entirely on the device through m3_track_gn_ray_dist; the host sees one
stream-ordered call and no per-iteration sync.

Poses are torch float32 tensors [8] or [1,8] = [t, q(xyzw), s] (Sim3.data layout,
liegroups/sim3.py:13).
"""
from __future__ import annotations

import torch

from . import _ffi
from .config import get_config


def _pose(T, name):
    T = _ffi.check(T.reshape(-1), torch.float32, name, (8,))
    return T


def _batch_of(t: torch.Tensor, inner: int) -> int:
    """Leading batch size of a [P,N,(inner)] / [N,(inner)] tensor (1 when unbatched)."""
    want = 3 if inner > 1 else 2
    return t.shape[0] if t.dim() == want else 1


def track_gather(Xf_canon, Cf_avg, Ck_avg, Qff, Qkf, idx_f2k, valid_match, C_conf=0.0, Q_conf=1.5):
    """tracker.py:88-113 + :214 fused: returns (Xf[idx] [N,3], Qk [N], valid_opt [N] u8,
    valid_kf [N] u8, counts int32[2] = (#valid_opt, #valid_kf)).
    Batched: Xf_canon [P,N,3] and idx_f2k [P,N] (all other inputs [P,N] or [P,N,1]) give [P,...] outputs
    from ONE launch."""
    batched = Xf_canon.dim() == 3 and idx_f2k.dim() == 2 and idx_f2k.shape[0] == Xf_canon.shape[0]
    P = Xf_canon.shape[0] if batched else 1
    n = idx_f2k.numel() // P
    Xf_canon = _ffi.check(Xf_canon.reshape(P, n, 3), torch.float32, "Xf_canon", (P, n, 3))
    Cf_avg = _ffi.check(Cf_avg.reshape(P, n), torch.float32, "Cf", (P, n))
    Ck_avg = _ffi.check(Ck_avg.reshape(P, n), torch.float32, "Ck", (P, n))
    Qff = _ffi.check(Qff.reshape(P, n), torch.float32, "Qff", (P, n))
    Qkf = _ffi.check(Qkf.reshape(P, n), torch.float32, "Qkf", (P, n))
    idx = _ffi.check(idx_f2k.reshape(P, n).to(torch.int64), torch.int64, "idx_f2k", (P, n))
    vm = _ffi.check(valid_match.reshape(P, n).to(torch.uint8), torch.uint8, "valid_match", (P, n))
    dev = Xf_canon.device
    Xf = torch.empty((P, n, 3), dtype=torch.float32, device=dev)
    Qk = torch.empty((P, n), dtype=torch.float32, device=dev)
    vo = torch.empty((P, n), dtype=torch.uint8, device=dev)
    vk = torch.empty((P, n), dtype=torch.uint8, device=dev)
    counts = torch.empty((P, 2), dtype=torch.int32, device=dev)
    _ffi.call("m3_track_gather_batch", _ffi.ptr(Xf_canon), _ffi.ptr(Cf_avg), _ffi.ptr(Ck_avg), _ffi.ptr(Qff),
              _ffi.ptr(Qkf), _ffi.ptr(idx), _ffi.ptr(vm), _ffi.ptr(Xf), _ffi.ptr(Qk), _ffi.ptr(vo), _ffi.ptr(vk),
              _ffi.ptr(counts), P, n, float(C_conf), float(Q_conf), _ffi.stream_ptr())
    if batched:
        return Xf, Qk, vo, vk, counts
    return Xf[0], Qk[0], vo[0], vk[0], counts[0]


def _ws(dev, P=1):
    return torch.empty(P * int(_ffi.lib().m3_track_ws_doubles()), dtype=torch.float64, device=dev)


def opt_pose_ray_dist_sim3(Xf, Xk, T_WCf, T_WCk, Qk, valid, cfg=None, fixed_iters: bool = False):
    """tracker.py:258-324.  Xf [N,3] (gathered at idx_f2k), Xk [N,3], Qk [N(,1)], valid [N(,1)].
    Returns (T_WCf [8], T_CkCf [8], info float64[4] = iterations, cost, |tau|, status: 0 budget used up /
    1 converged / 2 solve failed - singular or divergent, pose = the last good one).
    Batched: Xf [P,N,3] with poses [P,8] solves P independent problems in one launch sequence
    (outputs [P,8], [P,8], [P,4])."""
    c = dict(get_config()["tracking"])
    c.update(cfg or {})
    batched = Xf.dim() == 3
    P = Xf.shape[0] if batched else 1
    Xf = _ffi.check(Xf.reshape(P, -1, 3), torch.float32, "Xf")
    n = Xf.shape[1]
    Xk = _ffi.check(Xk.reshape(P, n, 3), torch.float32, "Xk", (P, n, 3))
    Qk = _ffi.check(Qk.reshape(P, n), torch.float32, "Qk", (P, n))
    v = _ffi.check(valid.reshape(P, n).to(torch.uint8), torch.uint8, "valid", (P, n))
    Tf = _ffi.check(T_WCf.reshape(-1, 8).expand(P, 8).contiguous(), torch.float32, "T_WCf", (P, 8))
    Tk = _ffi.check(T_WCk.reshape(-1, 8).expand(P, 8).contiguous(), torch.float32, "T_WCk", (P, 8))
    dev = Xf.device
    out_f = torch.empty((P, 8), dtype=torch.float32, device=dev)
    out_rel = torch.empty((P, 8), dtype=torch.float32, device=dev)
    info = torch.empty((P, 4), dtype=torch.float64, device=dev)
    ws = _ws(dev, P)
    _ffi.call("m3_track_gn_ray_dist_batch", _ffi.ptr(Xf), _ffi.ptr(Xk), _ffi.ptr(Qk), _ffi.ptr(v), _ffi.ptr(Tf),
              _ffi.ptr(Tk), _ffi.ptr(out_f), _ffi.ptr(out_rel), _ffi.ptr(info), _ffi.ptr(ws), P, n,
              int(c["max_iters"]), float(c["huber"]), float(c["sigma_ray"]), float(c["sigma_dist"]),
              float(c["rel_error"]), float(c["delta_norm"]), 1 if fixed_iters else 0, _ffi.stream_ptr())
    if batched:
        return out_f, out_rel, info
    return out_f[0], out_rel[0], info[0]


_K4_CACHE: dict = {}


def _k4(K):
    """[3,3] or (fx, fy, cx, cy) -> ctypes float[4] (host).  The intrinsics are launch PARAMETERS of the calibrated
    kernels; a device tensor is pulled to the host once per (storage, version) - not once per frame: with use_calib the
    tracker calls this three times per frame, each a host synchronisation in the first version."""
    import ctypes

    import numpy as np
    import weakref
    key = None
    if isinstance(K, torch.Tensor):
        key = id(K)
        hit = _K4_CACHE.get(key)
        if hit is not None and hit[0]() is K and hit[1] == K._version:      # the SAME tensor object, unmodified since
            return hit[2]
        Kh = K.detach().cpu().numpy()
    else:
        Kh = np.asarray(K)
    vals = (Kh[0, 0], Kh[1, 1], Kh[0, 2], Kh[1, 2]) if Kh.shape == (3, 3) else tuple(Kh.reshape(-1)[:4])
    out = (ctypes.c_float * 4)(*[float(v) for v in vals])
    if key is not None:
        if len(_K4_CACHE) > 64:
            _K4_CACHE.clear()
        _K4_CACHE[key] = (weakref.ref(K), K._version, out)
    return out


def constrain_points_to_ray(img_size, Xs, K):
    """geometry.py:273-302.  img_size = (height, width); Xs [P,H*W,3] or [H*W,3] -> same shape."""
    import ctypes
    h, w = img_size
    X = _ffi.check(Xs.reshape(-1, h * w, 3), torch.float32, "Xs")
    out = torch.empty_like(X)
    k4 = _k4(K)
    _ffi.call("m3_constrain_points_to_ray", _ffi.ptr(X), _ffi.ptr(out), X.shape[0], h, w,
              ctypes.cast(k4, ctypes.c_void_p), _ffi.stream_ptr())
    return out.reshape(Xs.shape)


def opt_pose_calib_sim3(Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, cfg=None, fixed_iters: bool = False):
    """tracker.py:326-406.  Xf [N,3] / [P,N,3] ray-constrained frame points gathered at idx_f2k, Xk the
    ray-constrained keyframe points in pixel order, K [3,3] or (fx,fy,cx,cy), img_size = (height, width).
    Returns (T_WCf, T_CkCf, info) like opt_pose_ray_dist_sim3."""
    import ctypes
    c = dict(get_config()["tracking"])
    c.setdefault("sigma_pixel", 1.0); c.setdefault("sigma_depth", 10.0)
    c.setdefault("pixel_border", 0); c.setdefault("depth_eps", 0.0)
    c.update(cfg or {})
    h, w = img_size
    batched = Xf.dim() == 3
    P = Xf.shape[0] if batched else 1
    n = h * w
    Xf = _ffi.check(Xf.reshape(P, n, 3), torch.float32, "Xf", (P, n, 3))
    Xk = _ffi.check(Xk.reshape(P, n, 3), torch.float32, "Xk", (P, n, 3))
    Qk = _ffi.check(Qk.reshape(P, n), torch.float32, "Qk", (P, n))
    v = _ffi.check(valid.reshape(P, n).to(torch.uint8), torch.uint8, "valid", (P, n))
    Tf = _ffi.check(T_WCf.reshape(-1, 8).expand(P, 8).contiguous(), torch.float32, "T_WCf", (P, 8))
    Tk = _ffi.check(T_WCk.reshape(-1, 8).expand(P, 8).contiguous(), torch.float32, "T_WCk", (P, 8))
    dev = Xf.device
    out_f = torch.empty((P, 8), dtype=torch.float32, device=dev)
    out_rel = torch.empty((P, 8), dtype=torch.float32, device=dev)
    info = torch.empty((P, 4), dtype=torch.float64, device=dev)
    ws = _ws(dev, P)
    k4 = _k4(K)
    _ffi.call("m3_track_gn_calib_batch", _ffi.ptr(Xf), _ffi.ptr(Xk), _ffi.ptr(Qk), _ffi.ptr(v), _ffi.ptr(Tf), _ffi.ptr(Tk),
              _ffi.ptr(out_f), _ffi.ptr(out_rel), _ffi.ptr(info), _ffi.ptr(ws), P, n, h, w,
              ctypes.cast(k4, ctypes.c_void_p), int(c["max_iters"]), float(c["huber"]), float(c["sigma_pixel"]),
              float(c["sigma_depth"]), float(c["pixel_border"]), float(c["depth_eps"]), float(c["rel_error"]),
              float(c["delta_norm"]), 1 if fixed_iters else 0, _ffi.stream_ptr())
    if batched:
        return out_f, out_rel, info
    return out_f[0], out_rel[0], info[0]


def normal_equations(Xf, Xk, T_CkCf, Qk, valid, cfg=None):
    """The J^T W J / J^T W r reduction of tracker.py:239-244 at pose T_CkCf.
    Returns (H [7,7] float64, g [7] float64, cost float64 scalar tensor)."""
    c = dict(get_config()["tracking"])
    c.update(cfg or {})
    Xf = _ffi.check(Xf.reshape(-1, 3), torch.float32, "Xf")
    n = Xf.shape[0]
    Xk = _ffi.check(Xk.reshape(-1, 3), torch.float32, "Xk", (n, 3))
    Qk = _ffi.check(Qk.reshape(-1), torch.float32, "Qk", (n,))
    v = _ffi.check(valid.reshape(-1).to(torch.uint8), torch.uint8, "valid", (n,))
    T = _pose(T_CkCf, "T_CkCf")
    dev = Xf.device
    out = torch.empty(36, dtype=torch.float64, device=dev)
    ws = _ws(dev)
    _ffi.call("m3_track_normal_eq", _ffi.ptr(Xf), _ffi.ptr(Xk), _ffi.ptr(Qk), _ffi.ptr(v), _ffi.ptr(T),
              _ffi.ptr(out), _ffi.ptr(ws), n, float(c["huber"]), float(c["sigma_ray"]), float(c["sigma_dist"]),
              _ffi.stream_ptr())
    iu = torch.triu_indices(7, 7, device=dev)
    H = torch.zeros((7, 7), dtype=torch.float64, device=dev)
    H[iu[0], iu[1]] = out[:28]
    H = H + H.T - torch.diag(H.diagonal())
    return H, out[28:35].clone(), out[35].clone()


def count_unique(idx: torch.Tensor, valid: torch.Tensor, value_range: int) -> torch.Tensor:
    """Number of distinct idx[n] with valid[n] (mx.unique(idx[valid]).shape[0], tracker.py:153-155) as a
    device int32 [1]: bitmap + popcount, no sort, no host sync."""
    idx = _ffi.check(idx.reshape(-1), torch.int64, "idx")
    valid = valid.reshape(-1)
    if valid.dtype == torch.bool:
        valid = valid.view(torch.uint8)
    valid = _ffi.check(valid, torch.uint8, "valid", (idx.numel(),))
    words = int(_ffi.lib().m3_count_unique_ws_words(value_range))
    ws = torch.empty(words, dtype=torch.int32, device=idx.device)
    out = torch.empty(1, dtype=torch.int32, device=idx.device)
    _ffi.call("m3_count_unique", _ffi.ptr(idx), _ffi.ptr(valid), idx.numel(), value_range, _ffi.ptr(ws), _ffi.ptr(out),
              _ffi.stream_ptr())
    return out


def sim3_act(T, X):
    """Sim3.act (liegroups/sim3.py:222-231) over a point map: s R X + t."""
    X = _ffi.check(X.reshape(-1, 3), torch.float32, "X")
    T = _pose(T, "T")
    out = torch.empty_like(X)
    _ffi.call("m3_sim3_act", _ffi.ptr(T), _ffi.ptr(X), _ffi.ptr(out), X.shape[0], _ffi.stream_ptr())
    return out


class FrameTracker:
    """tracker.py:21-175.  `keyframes` needs last_keyframe() and __len__/__setitem__; frames
    need X_canon [N,3], get_average_conf() [N,1], T_WC [1,8] tensor, update_pointmap(X, C),
    frame_id (see frame.py)."""

    def __init__(self, model, keyframes) -> None:
        self.model = model
        self.keyframes = keyframes
        self.cfg = get_config()["tracking"]
        self.idx_f2k = None
        self.last_info = None
        self.last_stats = None

    def reset_idx_f2k(self) -> None:
        self.idx_f2k = None

    def track(self, frame, mast3r_match_fn):
        """Returns (new_kf, match_info, try_reloc) as tracker.py:51-175."""
        keyframe = self.keyframes.last_keyframe()
        if keyframe is None:
            return False, [], True
        idx_f2k, valid_match_k, Xff, Cff, Qff, Xkf, Ckf, Qkf = mast3r_match_fn(
            self.model, frame, keyframe, idx_i2j_init=self.idx_f2k)
        self.idx_f2k = idx_f2k
        idx = idx_f2k[0]
        vm = valid_match_k[0].reshape(-1)
        n = idx.numel()
        frame.update_pointmap(Xff.reshape(n, 3), Cff.reshape(n, 1))
        use_calib = bool(get_config().get("use_calib", False)) and keyframe.K is not None
        img = frame.img
        img_size = (img.shape[1], img.shape[2]) if img.shape[0] == 3 else (img.shape[0], img.shape[1])
        Xf_canon, Xk_canon = frame.X_canon, keyframe.X_canon
        if use_calib:                                        # tracker.py:196-199
            Xf_canon = constrain_points_to_ray(img_size, Xf_canon, keyframe.K)
            Xk_canon = constrain_points_to_ray(img_size, Xk_canon, keyframe.K)
        Xf, Qk, valid_opt, valid_kf, counts = track_gather(
            Xf_canon, frame.get_average_conf(), keyframe.get_average_conf(), Qff, Qkf, idx, vm,
            self.cfg["C_conf"], self.cfg["Q_conf"])
        uniq = count_unique(idx, vm, n)          # tracker.py:153-155, needs only idx / valid_match
        # The solve is launched BEFORE the match-fraction gate is known: it only reads the gathered arrays and writes
        # its own outputs, so a frame the gate rejects (rare) costs 0.4 ms of discarded work - and the gate (:116), the
        # solve's failure flag (:139-141) and the keyframe statistics (:149-158) reach the host in ONE synchronisation.
        if use_calib:
            T_WCf, T_CkCf, info = opt_pose_calib_sim3(Xf, Xk_canon, frame.T_WC, keyframe.T_WC, Qk, valid_opt, keyframe.K,
                                                      img_size, self.cfg)
        else:
            T_WCf, T_CkCf, info = opt_pose_ray_dist_sim3(Xf, Xk_canon, frame.T_WC, keyframe.T_WC, Qk, valid_opt, self.cfg)
        host = torch.cat([counts.reshape(-1)[:2].double(), uniq.double(), info.reshape(-1)[3:4]]).cpu()
        self.last_info = info                                      # (iterations, cost, |tau|, status) of the last solve, on the device
        # The fractions below are "of the pixels" for the reference's dense matchers.  The fast reciprocal NN matcher
        # (matching.use_fast_nn, not in the reference) yields at most one match per seed: its fractions are of the SEEDS,
        # or every frame would fall below min_match_frac.
        from .matching import match_fraction_scale
        n = n / match_fraction_scale(img_size[0], img_size[1])     # the seed count with use_fast_nn, the pixel count otherwise
        if float(host[0]) / n < self.cfg["min_match_frac"]:
            print(f"Skipped frame {frame.frame_id}")
            return False, [], True
        if float(host[3]) == 2.0:                                  # the reference's `except` branch (tracker.py:139-141)
            print(f"Optimization failed for frame {frame.frame_id}: singular or divergent Gauss-Newton step")
            return False, [], True
        frame.T_WC = T_WCf.reshape(1, 8)
        # Xkk = T_CkCf.act(Xkf); keyframe.update_pointmap(Xkk, Ckf)  (tracker.py:146-147) in one kernel
        keyframe.update_pointmap(Xkf.reshape(-1, 3), Ckf.reshape(-1, 1), T=T_CkCf)
        self.keyframes[len(self.keyframes) - 1] = keyframe
        match_frac_k = float(host[1]) / n
        unique_frac_f = float(host[2]) / n
        self.last_stats = (match_frac_k, unique_frac_f)
        new_kf = min(match_frac_k, unique_frac_f) < self.cfg["match_frac_thresh"]
        if new_kf:
            self.reset_idx_f2k()
        match_info = [keyframe.X_canon, keyframe.get_average_conf(), frame.X_canon, frame.get_average_conf(),
                      Qkf, Qff]
        return new_kf, match_info, False
