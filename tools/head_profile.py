#!/usr/bin/env python3
"""Per-launch device time of ONE head (DPT + local features) at P = 8 pairs, 512x512: which MFMA launches and which
elementwise kernels the head's time goes to.  Eager, serialised; elementwise kernels are timed as the gaps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import model as M, ops
dev = torch.device("cuda:0")
net = M.Mast3rFull(seed=0, device=dev)
P, T = 8, 1024
g = torch.Generator().manual_seed(0)
taps = [torch.randn(P * T, c, generator=g).to(net.hdt).to(dev) for c in (1024, 768, 768, 768)]
for _ in range(2):
    net.head("downstream_head1", taps, P, (32, 32))
torch.cuda.synchronize()
ops.PROFILE, ops.PROFILE_SHAPES = [], []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); net.head("downstream_head1", taps, P, (32, 32)); e1.record(); torch.cuda.synchronize()
prof, shapes = ops.PROFILE, ops.PROFILE_SHAPES
ops.PROFILE = ops.PROFILE_SHAPES = None
tot = 0.0
print(f"head total (eager, instrumented): {e0.elapsed_time(e1):.3f} ms")
print("| launch | us | TFLOP/s |"); print("|---|---|---|")
for (kind, fl, a, b, nb), (_, desc) in zip(prof, shapes):
    us = a.elapsed_time(b) * 1e3; tot += us
    print(f"| {desc} | {us:.1f} | {fl / us / 1e6:.0f} |")
print(f"MFMA launches: {tot / 1e3:.3f} ms; the rest ({e0.elapsed_time(e1) - tot / 1e3:.3f} ms) is elementwise kernels and gaps")
