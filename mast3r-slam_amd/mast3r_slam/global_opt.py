"""Backend factor graph: builds the arrays the Gauss-Newton kernels consume.

Mirrors FactorGraph of /root/reference/src/mlx_mast3r_slam/global_opt.py (:14-270): add_factors
(:49-138), get_unique_kf_idx (:140-145), _prep_two_way_edges (:147-154), _get_poses_points (:156-166),
solve_GN_rays (:168-211), solve_GN_calib (:213-270).  Same names, arguments and bookkeeping; arrays are
torch tensors on the ROCm device; the solves call mast3r_slam.kernels.gauss_newton_rays / _calib (HIP:
per-edge blocks, assembly, Cholesky, retraction on the device) where the reference's MLX methods are
no-ops around a numpy/Metal kernel.

Two deliberate differences, both fixes of reference slips:
  * edges are re-indexed to positions in the unique-keyframe list before the solve, so Xs / poses (stacked
    over the UNIQUE keyframes, :156-166) and ii / jj always agree - the reference passes global ids;
  * solve_GN_calib hands the kernel img_size = (width, height), the order gauss_newton_calib unpacks
    (gauss_newton_calib.py:75); the reference passes (h, w) (global_opt.py:227).

Multi-GPU (BASELINE configs[4], no counterpart in the reference): FactorGraph(..., group=pg) shards the EDGES of
every add_factors call over the ranks of a torch.distributed group.  A rank decodes, matches, stores and linearises
only its own edges (the per-point arrays never travel); what is exchanged is one keep-flag per edge in add_factors
and 36 doubles per directed edge and Gauss-Newton iteration in the solve, after which every rank runs the same
deterministic device step - all ranks hold bit-identical poses.  `batch` bounds how many edges one call of the match
function sees (the symmetric decode of 2*batch pairs is the memory high-water mark).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import kernels, matching
from .config import get_config
from .tracker import constrain_points_to_ray


class FactorGraph:
    def __init__(self, model, frames, K: Optional[torch.Tensor] = None, group=None, batch: int = 8) -> None:
        self.model = model
        self.frames = frames
        self.K = K
        self.cfg = get_config()["local_opt"]
        self.group = group
        self.batch = max(int(batch), 1)
        self._dev = getattr(model, "device", "cuda")
        self.reset()

    def reset(self) -> None:
        """Drop every factor (a loop-closure re-match rebuilds the graph from the keyframe list)."""
        e = lambda dt: torch.empty((0,), dtype=dt, device=self._dev)
        self.ii, self.jj = e(torch.int32), e(torch.int32)   # ALL edges of the graph (every rank holds the full list)
        self.owner = e(torch.int32)                          # rank that stores edge k's matches (0 without a group)
        self._kept_per_rank: list[int] = []                  # host mirror of the owner counts (no sync when sizing the gather)
        self.idx_ii2jj = self.idx_jj2ii = None               # [E_own,N] int64 - this rank's edges only
        self.valid_match_j = self.valid_match_i = None       # [E_own,N,1] bool
        self.Q_ii2jj = self.Q_jj2ii = None                   # [E_own,N,1] float32

    def _rank_world(self):
        if self.group is None:
            return 0, 1
        import torch.distributed as tdist
        return tdist.get_rank(self.group), tdist.get_world_size(self.group)

    # ------------------------------------------------------------------ global_opt.py:49-138
    def add_factors(self, ii, jj, min_match_frac: float, mast3r_match_fn, is_reloc: bool = False) -> bool:
        ii, jj = [int(i) for i in ii], [int(j) for j in jj]
        rank, world = self._rank_world()
        from .dist import shard_range
        mine = shard_range(len(ii), rank, world)
        dev = self._dev
        cat = lambda xs: torch.cat([x if x.dim() == 3 else x[None] for x in xs])
        outs = []
        for lo in range(mine.start, mine.stop, self.batch):
            sel = range(lo, min(lo + self.batch, mine.stop))
            kf_ii = [self.frames[ii[k]] for k in sel]
            kf_jj = [self.frames[jj[k]] for k in sel]
            feat_i, feat_j = cat([k.feat for k in kf_ii]), cat([k.feat for k in kf_jj])
            pos_i, pos_j = cat([k.pos for k in kf_ii]), cat([k.pos for k in kf_jj])
            shape_i = [k.img_true_shape for k in kf_ii]
            shape_j = [k.img_true_shape for k in kf_jj]
            idx_i2j, idx_j2i, valid_match_j, valid_match_i, Qii, Qjj, Qji, Qij = mast3r_match_fn(
                self.model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j)
            Qj = torch.sqrt(torch.gather(Qii[..., 0], 1, idx_i2j)[..., None] * Qji)      # :92-93
            Qi = torch.sqrt(torch.gather(Qjj[..., 0], 1, idx_j2i)[..., None] * Qij)
            valid_j = valid_match_j.bool() & (Qj > self.cfg["Q_conf"])
            valid_i = valid_match_i.bool() & (Qi > self.cfg["Q_conf"])
            frac = torch.minimum(valid_j.float().mean(dim=(1, 2)), valid_i.float().mean(dim=(1, 2)))
            if matching.use_fast_nn_enabled():                                           # fractions of the SEEDS with use_fast_nn;
                hw = [int(v) for v in torch.as_tensor(shape_i[0]).reshape(-1)[:2]]       # the dense matchers (default) never read
                frac = frac * matching.match_fraction_scale(hw[0], hw[1])                # the shape tensor: no host sync per batch
            outs.append((idx_i2j, idx_j2i, valid_match_j.bool(), valid_match_i.bool(), Qj, Qi, frac))
        if outs:
            idx_i2j, idx_j2i, vmj, vmi, Qj, Qi, frac = (torch.cat([o[n] for o in outs]) for n in range(7))
        else:
            frac = torch.empty((0,), dtype=torch.float32, device=dev)
        if self.group is not None:
            from .dist import all_gather_rows
            frac = all_gather_rows(frac, len(ii), self.group)                            # one float per edge
        ii_t = torch.tensor(ii, dtype=torch.int32, device=dev)
        jj_t = torch.tensor(jj, dtype=torch.int32, device=dev)
        invalid = (~(ii_t == jj_t - 1)) & (frac < min_match_frac)                        # consecutive edges are always kept
        invalid_h = invalid.cpu()                                                        # one host sync per add_factors
        if bool(invalid_h.any()) and is_reloc:
            return False
        if int((~invalid_h).sum()) == 0:
            return False
        keep = ~invalid
        own_h = torch.cat([torch.full((len(shard_range(len(ii), r, world)),), r, dtype=torch.int32) for r in range(world)])
        kept_h = torch.bincount(own_h[~invalid_h].long(), minlength=world).tolist()      # invalid_h is on the host already
        if len(self._kept_per_rank) != world:
            self._kept_per_rank = [0] * world
        self._kept_per_rank = [a + b for a, b in zip(self._kept_per_rank, kept_h)]
        own = own_h.to(dev)
        self.ii, self.jj = torch.cat([self.ii, ii_t[keep]]), torch.cat([self.jj, jj_t[keep]])
        self.owner = torch.cat([self.owner, own[keep]])
        if outs:
            kl = keep[mine.start:mine.stop]
            app = lambda old, new: new[kl] if old is None else torch.cat([old, new[kl]])
            self.idx_ii2jj, self.idx_jj2ii = app(self.idx_ii2jj, idx_i2j), app(self.idx_jj2ii, idx_j2i)
            self.valid_match_j, self.valid_match_i = app(self.valid_match_j, vmj), app(self.valid_match_i, vmi)
            self.Q_ii2jj, self.Q_jj2ii = app(self.Q_ii2jj, Qj), app(self.Q_jj2ii, Qi)
        return True

    # ------------------------------------------------------------------ :140-166
    def get_unique_kf_idx(self) -> torch.Tensor:
        return torch.unique(torch.cat([self.ii, self.jj])).to(torch.int32)

    def _prep_two_way_edges(self):
        """global_opt.py:147-154 over the edges this rank stores (all of them without a group)."""
        rank, _ = self._rank_world()
        m = self.owner == rank
        ii = torch.cat([self.ii[m], self.jj[m]])
        jj = torch.cat([self.jj[m], self.ii[m]])
        if self.idx_ii2jj is None:
            n = self.frames[0].X_canon.shape[0]
            z = lambda dt, *tail: torch.empty((0, n) + tail, dtype=dt, device=self._dev)
            return ii, jj, z(torch.int64), z(torch.bool, 1), z(torch.float32, 1)
        idx = torch.cat([self.idx_ii2jj, self.idx_jj2ii])
        valid = torch.cat([self.valid_match_j, self.valid_match_i])
        Q = torch.cat([self.Q_ii2jj, self.Q_jj2ii])
        return ii, jj, idx, valid, Q

    def _get_poses_points(self, unique_kf_idx: torch.Tensor):
        kfs = [self.frames[int(i)] for i in unique_kf_idx.tolist()]
        Xs = torch.stack([k.X_canon for k in kfs])
        T_WCs = torch.stack([k.T_WC.reshape(8) for k in kfs])
        Cs = torch.stack([k.get_average_conf() for k in kfs])
        return Xs, T_WCs, Cs

    def _local_edges(self, unique_kf_idx):
        ii, jj, idx, valid, Q = self._prep_two_way_edges()
        u = unique_kf_idx.to(torch.int64)
        pos = lambda e: torch.searchsorted(u, e.to(torch.int64)).to(torch.int32)     # global keyframe id -> row of Xs
        graph = None
        if self.group is not None:
            # every rank's directed edges, rank by rank (forward then backward, as _prep_two_way_edges orders them):
            # the order the gathered 36-double blocks arrive in
            _, world = self._rank_world()
            gi, gj, sizes = [], [], []
            for r in range(world):
                m = self.owner == r
                gi += [self.ii[m], self.jj[m]]
                gj += [self.jj[m], self.ii[m]]
                sizes.append(2 * self._kept_per_rank[r])              # host-side count kept by add_factors: no device sync
            graph = (pos(torch.cat(gi)), pos(torch.cat(gj)), sizes)
        return pos(ii), pos(jj), idx.to(torch.int32), valid[..., 0], Q[..., 0], graph

    def _write_back(self, poses, unique_kf_idx, pin):
        upd = getattr(self.frames, "update_T_WCs", None)
        if upd is not None:
            upd(poses[pin:], unique_kf_idx[pin:])
            return
        for row, k in zip(poses[pin:], unique_kf_idx[pin:].tolist()):
            self.frames[int(k)].T_WC = row.reshape(1, 8).clone()

    # ------------------------------------------------------------------ :168-211
    def solve_GN_rays(self, max_iters: Optional[int] = None) -> None:
        pin = self.cfg["pin"]
        if self.ii.numel() == 0:
            return
        uniq = self.get_unique_kf_idx()
        if uniq.numel() <= pin:
            return
        Xs, T_WCs, Cs = self._get_poses_points(uniq)
        ii, jj, idx, valid, Q, graph = self._local_edges(uniq)
        poses = kernels.gauss_newton_rays(
            T_WCs, Xs, Cs, ii, jj, idx, valid, Q, sigma_ray=self.cfg["sigma_ray"], sigma_dist=self.cfg["sigma_dist"],
            C_thresh=self.cfg["C_conf"], Q_thresh=self.cfg["Q_conf"],
            max_iter=self.cfg["max_iters"] if max_iters is None else int(max_iters),
            delta_thresh=self.cfg["delta_norm"], pin=pin, group=self.group, graph=graph)
        self._write_back(poses, uniq, pin)

    # ------------------------------------------------------------------ :213-270
    def solve_GN_calib(self) -> None:
        if self.K is None:
            raise ValueError("Intrinsic matrix K required for calibrated mode")
        pin = self.cfg["pin"]
        if self.ii.numel() == 0:
            return
        uniq = self.get_unique_kf_idx()
        if uniq.numel() <= pin:
            return
        Xs, T_WCs, Cs = self._get_poses_points(uniq)
        img = self.frames[0].img
        h, w = (img.shape[1], img.shape[2]) if img.shape[0] == 3 else (img.shape[0], img.shape[1])
        Xs = constrain_points_to_ray((h, w), Xs, self.K)
        ii, jj, idx, valid, Q, graph = self._local_edges(uniq)
        poses = kernels.gauss_newton_calib(
            T_WCs, Xs, Cs, self.K, ii, jj, idx, valid, Q, (w, h), pixel_border=self.cfg.get("pixel_border", 0),
            z_eps=self.cfg.get("depth_eps", 0.0), sigma_pixel=self.cfg["sigma_pixel"], sigma_depth=self.cfg["sigma_depth"],
            C_thresh=self.cfg["C_conf"], Q_thresh=self.cfg["Q_conf"], max_iter=self.cfg["max_iters"],
            delta_thresh=self.cfg["delta_norm"], pin=pin, group=self.group, graph=graph)
        self._write_back(poses, uniq, pin)
