"""Minimal SLAM driver over the hot-path operators: the control flow of
/root/reference/src/mlx_mast3r_slam/slam.py (:124-153 main loop, :159-214 INIT / TRACKING,
:292-318 backend) with the retrieval database left out - a frame that cannot be tracked is
re-initialised as a new keyframe (the reference's "no similar keyframes" branch, :280-286).
It exists to show the operator API dropping in under the loop and to test it end to end; dataset
readers, trajectory writers and visualisation stay out of scope.
"""
from __future__ import annotations

from collections import deque
from typing import Callable, Iterable, Optional

import torch

from .config import get_config
from .frame import Keyframes, create_frame
from .global_opt import FactorGraph
from .mast3r_utils import mast3r_inference_mono, mast3r_match_asymmetric, mast3r_match_symmetric
from .tracker import FrameTracker, sim3_act

INIT, TRACKING, RELOC = "INIT", "TRACKING", "RELOC"


class SLAM:
    def __init__(self, model, K: Optional[torch.Tensor] = None) -> None:
        self.model = model
        self.config = get_config()
        self.keyframes = Keyframes()
        if K is not None:
            self.keyframes.set_intrinsics(K)
        self.tracker = FrameTracker(model, self.keyframes)
        self.factor_graph = FactorGraph(model, self.keyframes, K if self.config.get("use_calib") else None)
        self.mode = INIT
        self._queue: deque[int] = deque()
        self.timestamps: list = []
        self.poses: list[torch.Tensor] = []

    # ------------------------------------------------------------------ slam.py:124-153
    def run(self, frames: Iterable, callback: Optional[Callable] = None) -> dict:
        """frames: iterable of (timestamp, img) with img [3,H,W] float in [0,1] or uint8 [H,W,3]."""
        for i, (timestamp, img) in enumerate(frames):
            frame = create_frame(i, img if isinstance(img, torch.Tensor) else torch.as_tensor(img))
            frame.img = frame.img.to(self.model.device)
            frame.K = self.keyframes.get_intrinsics()
            if self.mode == INIT:
                self._process_init(frame)
            elif self.mode == TRACKING:
                self._process_tracking(frame)
            else:
                self._process_reloc(frame)
            self.timestamps.append(timestamp)
            self.poses.append(frame.T_WC)
            if callback:
                callback(frame, self.keyframes)
            self._run_backend()
        return self.results()

    def _mono(self, frame) -> None:
        X, C, feat, pos = mast3r_inference_mono(self.model, frame)
        frame.N, frame.N_updates = 0, 0
        frame.update_pointmap(X, C)                     # the frame owns its fusion buffers
        frame.feat, frame.pos = feat, pos

    def _add_keyframe(self, frame) -> None:
        self.keyframes.append(frame)
        self._queue.append(len(self.keyframes) - 1)

    def _process_init(self, frame) -> None:             # :159-182
        self._mono(frame)
        self._add_keyframe(frame)
        self.mode = TRACKING

    def _process_tracking(self, frame) -> None:         # :184-214
        new_kf, _, try_reloc = self.tracker.track(frame, mast3r_match_fn=mast3r_match_asymmetric)
        if try_reloc:
            self.mode = RELOC
            self._process_reloc(frame)
            return
        if new_kf:
            self._mono(frame)
            self._add_keyframe(frame)

    def _process_reloc(self, frame) -> None:            # :216-290 without the retrieval database
        self._mono(frame)
        last = self.keyframes.last_keyframe()
        if last is not None:
            frame.T_WC = last.T_WC.clone()
        self._add_keyframe(frame)
        self.mode = TRACKING
        self.tracker.reset_idx_f2k()

    def _run_backend(self) -> None:                     # :292-318
        while self._queue:
            idx = self._queue.popleft()
            if idx > 0:
                ii = list(range(max(0, idx - 3), idx))
                self.factor_graph.add_factors(ii, [idx] * len(ii),
                                              min_match_frac=self.config["local_opt"].get("min_match_frac", 0.1),
                                              mast3r_match_fn=mast3r_match_symmetric)
            if self.config.get("use_calib"):
                self.factor_graph.solve_GN_calib()
            else:
                self.factor_graph.solve_GN_rays()

    def results(self) -> dict:                          # :320-352 (tensors instead of numpy)
        pts = [sim3_act(kf.T_WC, kf.X_canon) for kf in self.keyframes._frames if kf.X_canon is not None]
        return {
            "timestamps": list(self.timestamps),
            "poses": torch.cat(self.poses) if self.poses else torch.empty((0, 8)),
            "points": torch.cat(pts) if pts else torch.empty((0, 3)),
            "keyframe_indices": [kf.frame_id for kf in self.keyframes._frames],
        }
