"""Level-1 kernel API: array-in / array-out operators on the MI355X.

Mirror of /root/reference/src/mlx_mast3r_slam/backends/mpsgraph/kernels.py (same names,
argument order and defaults: iter_proj :107, refine_matches :463, gauss_newton_rays :262,
is_available :27), backed by hand-written HIP through the C ABI (include/m3slam.h).

Inputs may be torch tensors on the ROCm device (zero-copy, results stay on the device)
or numpy arrays as in the reference (copied to the device and back; convenience only -
that path pays PCIe both ways).  There is no CPU implementation behind these functions.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _ffi

_DEV = "cuda"


def is_available() -> bool:
    """True when the HIP library is built and a ROCm device is visible."""
    import os
    return os.path.exists(_ffi.LIB_PATH) and torch.cuda.is_available()


def _to_dev(x, dtype):
    """numpy / tensor -> contiguous device tensor of `dtype`; returns (tensor, was_numpy)."""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x)).to(_DEV).to(dtype).contiguous(), True
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"expected numpy array or torch tensor, got {type(x).__name__}")
    if not x.is_cuda:
        raise RuntimeError("tensor inputs must live on the ROCm device; no CPU path exists")
    return x.to(dtype).contiguous(), False


def _out(t, as_numpy, np_dtype=None):
    if not as_numpy:
        return t
    a = t.cpu().numpy()
    return a.astype(np_dtype) if np_dtype is not None else a


# ------------------------------------------------------------------ iter_proj
def iter_proj(rays_with_grad, pts3d_norm, p_init, max_iter: int = 10, lambda_init: float = 1e-8,
              convergence_thresh: float = 1e-6, use_metal: bool = True, *, stop_scope: str = "global"):
    """kernels.py:107-148.  rays_with_grad [B,H,W,9], pts3d_norm [B,N,3], p_init [B,N,2]
    -> (p_final [B,N,2] float32, valid [B,N] bool).

    `use_metal` is accepted for signature compatibility and ignored.  stop_scope="global"
    reproduces the reference's early stop over the whole call; "batch" stops per batch item."""
    rays, np_in = _to_dev(rays_with_grad, torch.float32)
    tgt, _ = _to_dev(pts3d_norm, torch.float32)
    p0, _ = _to_dev(p_init, torch.float32)
    if rays.dim() != 4 or rays.shape[-1] != 9:
        raise ValueError(f"rays_with_grad must be [B,H,W,9], got {tuple(rays.shape)}")
    b, h, w, _ = rays.shape
    if tgt.dim() != 3 or tgt.shape[0] != b or tgt.shape[2] != 3:
        raise ValueError(f"pts3d_norm must be [B,N,3], got {tuple(tgt.shape)}")
    n = tgt.shape[1]
    if tuple(p0.shape) != (b, n, 2):
        raise ValueError(f"p_init must be [{b},{n},2], got {tuple(p0.shape)}")
    if stop_scope not in ("global", "batch"):
        raise ValueError("stop_scope must be 'global' or 'batch'")
    p_out = torch.empty_like(p0)
    valid = torch.empty((b, n), dtype=torch.uint8, device=rays.device)
    if n == 0:
        return _out(p_out, np_in), _out(valid.bool(), np_in)
    ws = torch.empty(int(_ffi.lib().m3_iter_proj_ws_words(b, n, max(int(max_iter), 0))), dtype=torch.int32,
                     device=rays.device)
    _ffi.call("m3_iter_proj", _ffi.ptr(rays), _ffi.ptr(tgt), _ffi.ptr(p0), _ffi.ptr(p_out), _ffi.ptr(valid),
              _ffi.ptr(ws), b, h, w, n, int(max_iter), float(lambda_init), float(convergence_thresh),
              0 if stop_scope == "global" else 1, _ffi.stream_ptr())
    return _out(p_out, np_in), _out(valid.bool(), np_in)


# ------------------------------------------------------------------ refine_matches
def refine_matches(D11, D21, p1, radius: int = 3, dilation_max: int = 0, use_metal: bool = True, *,
                   chained: bool = False):
    """kernels.py:463-493.  D11 [B,H,W,D], D21 [B,N,D], p1 [B,N,2] (int; floats are truncated)
    -> refined int32 [B,N,2].  chained=False: numpy-twin semantics; True: Metal semantics.
    torch.float16 descriptors (both arrays) take the half-storage kernel: same fp32 scoring on the widened values."""
    half = isinstance(D11, torch.Tensor) and D11.dtype == torch.float16
    if half and not (isinstance(D21, torch.Tensor) and D21.dtype == torch.float16):
        raise ValueError("D11 is float16: D21 must be float16 as well")
    d11, np_in = _to_dev(D11, torch.float16 if half else torch.float32)
    d21, _ = _to_dev(D21, torch.float16 if half else torch.float32)
    if isinstance(p1, np.ndarray):
        p1 = torch.from_numpy(np.ascontiguousarray(p1)).to(_DEV)
    if p1.dtype.is_floating_point:
        p1 = p1.to(torch.float32).trunc()
    p = p1.to(torch.int32).contiguous()
    if d11.dim() != 4:
        raise ValueError(f"D11 must be [B,H,W,D], got {tuple(d11.shape)}")
    b, h, w, d = d11.shape
    if d21.dim() != 3 or d21.shape[0] != b or d21.shape[2] != d:
        raise ValueError(f"D21 must be [B,N,{d}], got {tuple(d21.shape)}")
    n = d21.shape[1]
    if tuple(p.shape) != (b, n, 2):
        raise ValueError(f"p1 must be [{b},{n},2], got {tuple(p.shape)}")
    out = torch.empty_like(p)
    if n == 0:
        return _out(out, np_in)
    _ffi.call("m3_refine_matches_f16" if half else "m3_refine_matches", _ffi.ptr(d11), _ffi.ptr(d21), _ffi.ptr(p),
              _ffi.ptr(out), b, h, w, d, n, int(radius), int(dilation_max), 1 if chained else 0, _ffi.stream_ptr())
    return _out(out, np_in)


# ------------------------------------------------------------------ gauss_newton_rays
def _local_map(ii, jj, num_kf: int, pin: int):
    """gauss_newton.py:66-81: unique keyframes, first `pin` of them fixed."""
    ii_h = ii.cpu().numpy() if isinstance(ii, torch.Tensor) else np.asarray(ii)
    jj_h = jj.cpu().numpy() if isinstance(jj, torch.Tensor) else np.asarray(jj)
    uniq = np.unique(np.concatenate([ii_h, jj_h]))
    local = np.full(num_kf, -1, dtype=np.int32)
    for i, kf in enumerate(uniq):
        if 0 <= kf < num_kf:
            local[int(kf)] = i - pin if i >= pin else -1
    return uniq, local, max(len(uniq) - pin, 0)


def _calib_ptr(calib):
    """10 host floats -> ctypes float[10] (the C side copies it during the call; ctypes passes the
    array by reference and keeps it alive for the duration of the call)."""
    if calib is None:
        return None
    import ctypes
    return (ctypes.c_float * 10)(*[float(v) for v in calib])


def gn_rays_blocks(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_ray: float = 0.003,
                   C_thresh: float = 0.0, Q_thresh: float = 1.5, point_mode: bool = False):
    """Per-edge normal-equation blocks [E,36] float64 = (Hjj upper 28, gj 7, count)."""
    t = _prep_gn(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q)
    k, p, e = t["K"], t["P"], t["E"]
    chunks = _ffi.lib().m3_gn_rays_chunks(p)
    blocks = torch.empty((e, 36), dtype=torch.float64, device=t["Twc"].device)
    ws = torch.empty(e * chunks * 36, dtype=torch.float64, device=t["Twc"].device)
    _ffi.call("m3_gn_rays_blocks", _ffi.ptr(t["Twc"]), _ffi.ptr(t["Xs"]), _ffi.ptr(t["Cs"]), _ffi.ptr(t["ii"]),
              _ffi.ptr(t["jj"]), _ffi.ptr(t["idx"]), _ffi.ptr(t["valid"]), _ffi.ptr(t["Q"]), _ffi.ptr(blocks),
              _ffi.ptr(ws), k, p, e, float(sigma_ray), float(C_thresh), float(Q_thresh), int(point_mode), None,
              _ffi.stream_ptr())
    return _out(blocks, t["np_in"])


def _prep_gn(Twc, Xs, Cs, ii, jj, idx, valid, Q):
    twc, np_in = _to_dev(Twc, torch.float32)
    xs, _ = _to_dev(Xs, torch.float32)
    cs, _ = _to_dev(Cs, torch.float32)
    q, _ = _to_dev(Q, torch.float32)
    if isinstance(valid, np.ndarray):
        valid = torch.from_numpy(np.ascontiguousarray(valid)).to(_DEV)
    vm = valid.to(torch.uint8).contiguous()
    ii_d, _ = _to_dev(ii, torch.int32)
    jj_d, _ = _to_dev(jj, torch.int32)
    idx_d, _ = _to_dev(idx, torch.int32)
    if cs.dim() == 3:
        cs = cs[..., 0].contiguous()
    if q.dim() == 3:
        q = q[..., 0].contiguous()
    if vm.dim() == 3:
        vm = vm[..., 0].contiguous()
    if twc.dim() != 2 or twc.shape[1] != 8:
        raise ValueError(f"Twc must be [K,8], got {tuple(twc.shape)}")
    k = twc.shape[0]
    if xs.dim() != 3 or xs.shape[0] != k or xs.shape[2] != 3:
        raise ValueError(f"Xs must be [K,P,3], got {tuple(xs.shape)}")
    p = xs.shape[1]
    e = ii_d.numel()
    for name, t, shp in (("Cs", cs, (k, p)), ("idx_ii2jj", idx_d, (e, p)), ("valid_match", vm, (e, p)),
                         ("Q", q, (e, p)), ("jj", jj_d, (e,))):
        if tuple(t.shape) != shp:
            raise ValueError(f"{name} must have shape {shp}, got {tuple(t.shape)}")
    return dict(Twc=twc, Xs=xs, Cs=cs, ii=ii_d, jj=jj_d, idx=idx_d, valid=vm, Q=q, K=k, P=p, E=e, np_in=np_in)


def gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_ray: float = 0.003,
                      sigma_dist: float = 10.0, C_thresh: float = 0.0, Q_thresh: float = 1.5,
                      max_iter: int = 10, delta_thresh: float = 1e-4, pin: int = 1, use_metal: bool = True,
                      *, return_info: bool = False, _point_mode: int = 0, _calib=None, group=None, graph=None):
    """kernels.py:262-322 / gauss_newton.py:23-280.  Returns updated Twc [K,8] float32
    (input is not modified).  sigma_dist is accepted and ignored, as in the reference.

    group: a torch.distributed process group -> the edges are split over its ranks: each rank evaluates the
    per-point blocks of its own edges (10.8 MB of reads per edge), the 36-double blocks are all-gathered (288 B per
    edge) and EVERY rank runs the same deterministic device step on the gathered blocks (fixed-order assembly,
    Cholesky, stop test, retraction): bit-identical poses on all ranks, no broadcast.  With the nccl (RCCL) backend
    nothing inside the loop synchronises with the host (the edge list is read to the host ONCE before it, to build the
    keyframe map); with gloo (CPU tests, several ranks on one GPU) the blocks travel through host memory every iteration.
      * graph=None: every rank passes the FULL graph and takes its shard_range() slice of the edges;
      * graph=(ii_all, jj_all, sizes): the edge arguments are this rank's OWN directed edges only (sizes[rank]
        of them - the matches never leave the rank that computed them, BASELINE configs[4]) and ii_all / jj_all
        list all sum(sizes) directed edges in rank order (rank 0's edges first)."""
    num_kf = Twc.shape[0]
    as_np = isinstance(Twc, np.ndarray)
    sharded = group is not None
    if graph is not None and not sharded:
        raise ValueError("graph=(ii_all, jj_all, sizes) needs a process group")
    ii_g, jj_g = (graph[0], graph[1]) if graph is not None else (ii, jj)
    num_edges = len(ii_g)

    def _unchanged():
        out = Twc.copy() if as_np else Twc.clone()
        return (out, dict(iters=0, last_dx=0.0, stopped=True, failed=False)) if return_info else out
    if num_edges == 0 or num_kf <= pin:
        return _unchanged()
    uniq, local_h, num_free = _local_map(ii_g, jj_g, num_kf, pin)
    if len(uniq) <= pin or num_free <= 0:
        return _unchanged()
    t = _prep_gn(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q)
    k, p, e = t["K"], t["P"], t["E"]
    dev = t["Twc"].device
    twc = t["Twc"].clone()
    local = torch.from_numpy(local_h).to(dev)
    L = _ffi.lib()
    chunks = L.m3_gn_rays_chunks(p)
    dim = 7 * num_free
    info = torch.zeros(4, dtype=torch.float64, device=dev)
    st = _ffi.stream_ptr()
    hbuf = torch.empty(int(L.m3_gn_rays_hbuf_doubles(dim)), dtype=torch.float64, device=dev)
    if not sharded:
        blocks = torch.empty((e, 36), dtype=torch.float64, device=dev)
        ws = torch.empty(e * chunks * 36, dtype=torch.float64, device=dev)
        # the whole loop (blocks -> assemble -> Cholesky of ANY size -> stop test -> retract) in one stream-ordered call
        _ffi.call("m3_gn_rays_solve", _ffi.ptr(twc), _ffi.ptr(t["Xs"]), _ffi.ptr(t["Cs"]), _ffi.ptr(t["ii"]),
                  _ffi.ptr(t["jj"]), _ffi.ptr(t["idx"]), _ffi.ptr(t["valid"]), _ffi.ptr(t["Q"]), _ffi.ptr(local),
                  _ffi.ptr(blocks), _ffi.ptr(ws), _ffi.ptr(hbuf), _ffi.ptr(info), k, p, e, num_free,
                  float(sigma_ray), float(C_thresh), float(Q_thresh), int(max_iter), float(delta_thresh),
                  int(_point_mode), _calib_ptr(_calib), st)
    else:
        import torch.distributed as tdist
        from . import dist as m3dist
        rank, world = tdist.get_rank(group), tdist.get_world_size(group)
        if graph is None:
            mine = m3dist.shard_range(e, rank, world)
            sl = slice(mine.start, mine.stop)
            loc = {n: t[n][sl].contiguous() for n in ("ii", "jj", "idx", "valid", "Q")}
            sizes = [len(m3dist.shard_range(e, r, world)) for r in range(world)]
            ii_all, jj_all = t["ii"], t["jj"]
        else:
            loc = {n: t[n] for n in ("ii", "jj", "idx", "valid", "Q")}
            sizes = [int(x) for x in graph[2]]
            if len(sizes) != world or sizes[rank] != e or sum(sizes) != num_edges:
                raise ValueError(f"graph sizes {sizes} do not match {e} local / {num_edges} global edges on rank {rank}")
            ii_all, _ = _to_dev(ii_g, torch.int32)
            jj_all, _ = _to_dev(jj_g, torch.int32)
        e_loc, e_all = sizes[rank], sum(sizes)
        blocks_loc = torch.empty((e_loc, 36), dtype=torch.float64, device=dev)
        ws = torch.empty(max(e_loc, 1) * chunks * 36, dtype=torch.float64, device=dev)
        # The stop / failure flags live on the device (info); a stopped solve turns the remaining launches into no-ops.
        _ffi.call("m3_gn_rays_info_init", _ffi.ptr(info), st)
        for _ in range(max_iter):
            if e_loc:
                _ffi.call("m3_gn_rays_blocks", _ffi.ptr(twc), _ffi.ptr(t["Xs"]), _ffi.ptr(t["Cs"]), _ffi.ptr(loc["ii"]),
                          _ffi.ptr(loc["jj"]), _ffi.ptr(loc["idx"]), _ffi.ptr(loc["valid"]), _ffi.ptr(loc["Q"]),
                          _ffi.ptr(blocks_loc), _ffi.ptr(ws), k, p, e_loc, float(sigma_ray), float(C_thresh),
                          float(Q_thresh), int(_point_mode), _calib_ptr(_calib), st)
            blocks = m3dist.all_gather_rows(blocks_loc, e_all, group, sizes).contiguous()
            _ffi.call("m3_gn_rays_step", _ffi.ptr(twc), _ffi.ptr(blocks), _ffi.ptr(ii_all), _ffi.ptr(jj_all),
                      _ffi.ptr(local), _ffi.ptr(hbuf), _ffi.ptr(info), k, e_all, num_free, float(delta_thresh), st)
    result_info = None
    if return_info:
        i = info.cpu().numpy()
        result_info = dict(iters=int(i[0]), last_dx=float(i[1]), stopped=bool(i[2]), failed=bool(i[3]))
    out = _out(twc, t["np_in"], np.float32)
    return (out, result_info) if return_info else out


def cholesky_solve(H, g, reg: float = 1e-6):
    """backends/mpsgraph/linalg.py:17-50 on the device: x = solve(H + reg I, g) for a symmetric positive-definite H
    [N,N] (or batched [B,N,N]) of ANY size by the blocked float64 Cholesky of gn_chol.hip.  numpy in -> numpy out,
    tensors in -> tensor out; the arithmetic is float64, the result has H's dtype (as the reference: float32 in ->
    float32 out).

    Difference from the reference, on purpose: the reference calls LAPACK's LU (any non-singular H) and falls back
    to lstsq on a singular one; its contract, and every caller on the hot path, is a Gauss-Newton normal matrix
    (SPD after + reg I).  H + reg I that is NOT positive definite raises RuntimeError here instead of returning an
    LU / least-squares answer - no silent downgrade.  A batched call checks the status flags of all items with ONE
    host synchronisation after the last launch."""
    if isinstance(H, np.ndarray):
        out_dtype = torch.from_numpy(np.empty(0, dtype=H.dtype)).dtype if H.dtype in (np.float32, np.float64) else torch.float64
    else:
        out_dtype = H.dtype if isinstance(H, torch.Tensor) and H.dtype.is_floating_point else torch.float64
    Hd, np_in = _to_dev(H, torch.float64)
    gd, _ = _to_dev(g, torch.float64)
    batched = Hd.dim() == 3
    Hb = Hd if batched else Hd[None]
    nb, n = Hb.shape[0], Hb.shape[1]
    if Hb.dim() != 3 or Hb.shape[2] != n or gd.numel() != nb * n:
        raise ValueError(f"H must be [N,N] / [B,N,N] and g [N] / [B,N], got {tuple(Hd.shape)} / {tuple(gd.shape)}")
    Hw, bw = Hb.clone().contiguous(), gd.reshape(nb, n).clone().contiguous()
    x = torch.empty((nb, n), dtype=torch.float64, device=Hw.device)
    wsn = int(_ffi.lib().m3_chol_ws_doubles(n))
    ws = torch.empty((nb, wsn), dtype=torch.float64, device=Hw.device)
    for i in range(nb):
        _ffi.call("m3_chol_solve", _ffi.ptr(Hw[i]), _ffi.ptr(bw[i]), _ffi.ptr(x[i]), _ffi.ptr(ws[i]), n, float(reg),
                  _ffi.stream_ptr())
    bad = torch.nonzero(ws[:, 0]).reshape(-1).tolist()                  # the one host synchronisation
    if bad:
        raise RuntimeError(f"cholesky_solve: H + reg*I is not positive definite (batch items {bad})")
    x = x.to(out_dtype)
    return _out(x if batched else x[0], np_in)


def gauss_newton_points(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_point: float = 0.01,
                        C_thresh: float = 0.0, Q_thresh: float = 1.5, max_iter: int = 10,
                        delta_thresh: float = 1e-4, pin: int = 1, use_metal: bool = True, *, return_info: bool = False):
    """kernels.py:396-460 / gauss_newton_points.py:17-207: the 3-D point alignment with the extra
    scale-invariant weight 1/(|Xi| + 1e-6).  Same device path as gauss_newton_rays."""
    return gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_ray=sigma_point,
                             C_thresh=C_thresh, Q_thresh=Q_thresh, max_iter=max_iter, delta_thresh=delta_thresh,
                             pin=pin, return_info=return_info, _point_mode=1)


def gauss_newton_calib(Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, img_size, pixel_border: int = 0,
                       z_eps: float = 0.0, sigma_pixel: float = 1.0, sigma_depth: float = 0.1, C_thresh: float = 0.0,
                       Q_thresh: float = 1.5, max_iter: int = 10, delta_thresh: float = 1e-4, pin: int = 1,
                       use_metal: bool = True, *, return_info: bool = False, group=None, graph=None):
    """kernels.py:325-393 / gauss_newton_calib.py:17-274: calibrated projection residual
    ((du, dv)/sigma_pixel, dlog z/sigma_depth).  K is [3,3] or (fx, fy, cx, cy); img_size = (width, height)."""
    Kh = K.cpu().numpy() if isinstance(K, torch.Tensor) else np.asarray(K)
    fx, fy, cx, cy = (Kh[0, 0], Kh[1, 1], Kh[0, 2], Kh[1, 2]) if Kh.shape == (3, 3) else Kh.flatten()[:4]
    w, h = img_size
    calib = (fx, fy, cx, cy, w, h, pixel_border, z_eps, sigma_pixel, sigma_depth)
    return gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_ray=1.0, C_thresh=C_thresh,
                             Q_thresh=Q_thresh, max_iter=max_iter, delta_thresh=delta_thresh, pin=pin,
                             return_info=return_info, _point_mode=2, _calib=calib, group=group, graph=graph)
