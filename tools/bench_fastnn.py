#!/usr/bin/env python3
"""Fast reciprocal NN (K8) in isolation: P pairs of 512 x 512 fp16 / fp32 descriptor maps, 64 x 64 seeds, 3 rounds.
Prints the device time of one m3_frnn_round (two MFMA searches + bookkeeping) and the MFMA rate of a search."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import numpy as np
import torch
from mast3r_slam import _ffi, matching, synthetic
P = int(sys.argv[sys.argv.index("--pairs") + 1]) if "--pairs" in sys.argv else 8
dev = torch.device("cuda:0")
H = W = 512
sc = synthetic.geometric_pair(H, W, seed=1000, batch=P)
D1 = torch.from_numpy(sc["D21"]).to(dev)
D2 = torch.from_numpy(sc["D11"]).to(dev)
for half in (True, False):
    a, b = (D1.half(), D2.half()) if half else (D1, D2)
    for _ in range(2):
        matching.fast_reciprocal_nn_device(a, b, subsample=8, max_iter=3)
    torch.cuda.synchronize()
    _ffi.PROFILE, _ffi.PROFILE_NAMES = {}, ("m3_frnn_round",)
    for _ in range(3):
        out = matching.fast_reciprocal_nn_device(a, b, subsample=8, max_iter=3)
    torch.cuda.synchronize()
    evs, _ffi.PROFILE = _ffi.PROFILE["m3_frnn_round"], None
    us = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
    med = us[len(us) // 2]
    fl = 2.0 * P * 4096 * H * W * 24
    print(f"{'fp16' if half else 'fp32'} descriptors, {P} pairs: m3_frnn_round median {med:.0f} us (min {us[0]:.0f}) -> "
          f"{2 * fl / med / 1e6:.0f} TFLOP/s per search (D = 24), {out[-1].numel()} reciprocal pairs")
