"""Hot-path constants (mirror of DEFAULT_CONFIG, /root/reference/src/mlx_mast3r_slam/config.py:55-114).

Only the keys the hot path reads are kept.  `matching.py:405-407` of the reference reads
`refine_radius` / `refine_dilation` / `use_refine`, which are absent from its defaults and
therefore resolve to 3 / 2 / True; they are explicit here.
"""
from __future__ import annotations

import copy
from typing import Any

DEFAULT_CONFIG: dict[str, Any] = {
    "use_calib": False,
    "dataset": {"img_size": 512, "img_downsample": 1},
    "matching": {
        "use_simple": True,
        "max_iter": 10,
        "lambda_init": 1e-8,
        "convergence_thresh": 1e-6,
        "dist_thresh": 0.1,
        "radius": 3,
        "dilation_max": 0,
        "refine_radius": 3,
        "refine_dilation": 2,
        "use_refine": True,
        "refine_chained": False,     # False = numpy-twin semantics, True = Metal/CUDA-original
        # not in the reference (it has no fast reciprocal NN; BASELINE.json's north_star names it): when set, matching.match
        # dispatches to match_fast_nn - seeds every `fast_nn_subsample` pixels, `fast_nn_rounds` forward / backward rounds
        "use_fast_nn": False,
        "fast_nn_subsample": 8,
        "fast_nn_rounds": 3,
    },
    "tracking": {
        "min_match_frac": 0.05,
        "C_conf": 0.0,
        "Q_conf": 1.5,
        "rel_error": 1e-3,
        "delta_norm": 1e-3,
        "max_iters": 10,
        "huber": 1.345,
        "sigma_ray": 0.003,
        "sigma_dist": 10.0,
        "sigma_pixel": 1.0,
        "sigma_depth": 10.0,
        "pixel_border": 0,
        "depth_eps": 0.0,
        "match_frac_thresh": 0.333,
        "filtering_mode": "weighted_pointmap",      # config.py:89-90 of the reference
        "filtering_score": "median",
    },
    "local_opt": {
        "window_size": 1000000,
        "pin": 1,
        "max_iters": 10,
        "C_conf": 0.0,
        "Q_conf": 1.5,
        "sigma_ray": 0.003,
        "sigma_dist": 10.0,
        "sigma_pixel": 1.0,
        "sigma_depth": 10.0,
        "pixel_border": 0,
        "depth_eps": 0.0,
        "delta_norm": 1e-3,
        "min_match_frac": 0.1,
    },
    "reloc": {
        "min_match_frac": 0.3,
        "strict": True,
    },
}

config: dict[str, Any] = {}


def get_config() -> dict[str, Any]:
    """config.py:117-121: the loaded config, else a copy of the defaults."""
    if not config:
        return copy.deepcopy(DEFAULT_CONFIG)
    return config


def set_config(new: dict[str, Any]) -> None:
    """Deep-merge `new` over the defaults and install it as the active config."""
    merged = copy.deepcopy(DEFAULT_CONFIG)

    def _merge(base, upd):
        for k, v in upd.items():
            if isinstance(v, dict) and isinstance(base.get(k), dict):
                _merge(base[k], v)
            else:
                base[k] = v
    _merge(merged, new)
    config.clear()
    config.update(merged)


def reset_config() -> None:
    config.clear()
