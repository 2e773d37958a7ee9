#!/usr/bin/env python3
"""The matcher kernels of one benchmark step in isolation (8 maps of 512 x 512 on the smooth synthetic two-view scene,
fp32 and fp16 descriptors): prep -> iter_proj -> refine_matches -> epilogue through the C ABI, HIP-event time per call.
Run it under `rocprofv3 --pmc ...` (program directly after `--`) for the SQ counters of k_iter_proj / k_refine_lds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import numpy as np
import torch
from mast3r_slam import _ffi, config, matching, synthetic
P = int(sys.argv[sys.argv.index("--maps") + 1]) if "--maps" in sys.argv else 8
REPS = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 10
dev = torch.device("cuda:0")
H = W = 512
config.set_config({"matching": {"use_simple": False}})
sc = synthetic.geometric_pair(H, W, seed=1000, batch=P)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
X11, X21, D11, D21 = t(sc["X11"]), t(sc["X21"]), t(sc["D11"]), t(sc["D21"])
names = ("m3_prep_iter_proj", "m3_iter_proj", "m3_refine_matches", "m3_refine_matches_f16", "m3_match_epilogue")
for half in (False, True):
    a, b = (D11.half(), D21.half()) if half else (D11, D21)
    for _ in range(2):
        matching.match(X11, X21, a, b)
    torch.cuda.synchronize()
    _ffi.PROFILE, _ffi.PROFILE_NAMES = {}, names
    for _ in range(REPS):
        idx, valid = matching.match(X11, X21, a, b)
    torch.cuda.synchronize()
    prof, _ffi.PROFILE = _ffi.PROFILE, None
    print(f"descriptors {'fp16' if half else 'fp32'}: valid fraction {float(valid.float().mean()):.3f}")
    for n in names:
        ev = prof.get(n)
        if ev:
            us = sorted(x.elapsed_time(y) * 1e3 for x, y in ev)
            print(f"  {n:26s} median {us[len(us) // 2]:8.1f} us   min {us[0]:8.1f} us   ({P} maps of {H}x{W})")
